#!/bin/bash
# pmc_quick.sh TAG "BENCH ARGS" COUNTERS...: one counter pass over a short bench run; per-kernel totals on stdout
tag=$1; args=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-count-pass > $out.json 2> $out.err || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $out
