#!/bin/bash
# pmc_run.sh NAME "BENCH ARGS" COUNTERS...: one rocprofv3 counter pass over a short bench run with the given arguments
# (environment knobs are taken from the caller's environment); csv lands in gpurun_out/pmc_NAME/
name=$1; args=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-count-pass > $out.log 2>&1
