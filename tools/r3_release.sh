#!/bin/bash
# r3_release.sh: everything the profiles/ directory records for one build (run on the GPU box): GPU tests, the bench line
# with its kernel trace and HBM traffic counters, the other workloads, counter tables of the three mesh workloads, build
# times of the two tree builders.  Results under gpurun_out/rel/.
set -o pipefail
cd $GRAFT_REPO_ROOT
out=gpurun_out/rel
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -5 $out/tests.log; exit 1; }
tail -1 $out/tests.log
bash tools/profile_bench.sh rel > $out/profile_bench.log 2>&1 || exit 1
p=gpurun_out/prof_rel
python3 tools/pmc_summary.py $(find $p/fetch -name "*counter_collection.csv" | head -1) $(find $p/write -name "*counter_collection.csv" | head -1) c3_utah $out/pmc_traffic.json > $out/pmc_summary.log 2>&1 || exit 1
bash tools/other_workloads.sh > $out/other_workloads.txt 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --workload c4_miku --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $out/c4_64spp.json 2> $out/c4.err || exit 1
timeout -k 10 200 python3 bench.py --workload c5_composite --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > $out/c5_16spp.json 2> $out/c5.err || exit 1
bash tools/pmc_workload.sh rel_c3 "--workload c3_utah" > $out/pmc_c3.log 2>&1 || exit 1
bash tools/pmc_workload.sh rel_c4 "--workload c4_miku --spp 64" > $out/pmc_c4.log 2>&1 || exit 1
bash tools/pmc_workload.sh rel_c5 "--workload c5_composite --spp 8" > $out/pmc_c5.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/bvh_build_times.py utah miku composite > $out/bvh_build_times.txt 2>/dev/null || exit 1

timeout -k 10 200 python3 tools/fixed_cost.py > $out/fixed_cost.txt 2>/dev/null || exit 1
echo release done
