#!/bin/bash
# r3_collect.sh: copies what tools/r3_release.sh left under gpurun_out/ into profiles/ (run in the repository, after the GPU call)
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/rel/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/rel/pmc_traffic.json profiles/r3_pmc_traffic.json
cp gpurun_out/prof_rel/bench_under_rocprof.json profiles/r3_bench_under_rocprof.json
cp "$(find gpurun_out/prof_rel/trace -name '*kernel_stats.csv' | head -1)" profiles/r3_bench_kernel_stats.csv
cp gpurun_out/rel/bvh_build_times.txt profiles/r3_bvh_build_times.txt
cp gpurun_out/rel/c4_64spp.json profiles/r3_c4_64spp.json
cp gpurun_out/rel/c5_16spp.json profiles/r3_c5_16spp.json
cp gpurun_out/rel/fixed_cost.txt profiles/r3_fixed_cost.txt
cp gpurun_out/rel/other_workloads.txt profiles/r3_other_workloads.txt
for w in c3 c4 c5; do
  cp "$(find gpurun_out/pmc_rel_$w/trace -name '*kernel_stats.csv' | head -1)" profiles/r3_${w}_kernel_stats.csv
  python3 tools/pmc_report.py gpurun_out/pmc_rel_$w "Hardware counters, workload $w" profiles/r3_${w}_counters.md profiles/r3_${w}_counters.json > /dev/null
done
