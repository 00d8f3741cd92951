#!/bin/bash
# profile_bench.sh TAG: full bench line, rocprofv3 kernel-trace stats and the two HBM traffic counter passes of the
# headline workload (run on the GPU box; results under gpurun_out/prof_TAG/)
set -o pipefail
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-count-pass > $out/bench_under_rocprof.json 2> $out/trace.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-count-pass > $out/fetch.json 2> $out/fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-count-pass > $out/write.json 2> $out/write.err || exit 1
find $out -name "*.csv" | head -20
