#!/bin/bash
# r3_profiles.sh: kernel traces and counter passes of the three mesh workloads, the per-lane load micro-benchmark at the
# Miku scene's footprint, and the full bench lines of C4 / C5 at reduced sample counts (run on the GPU box)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
timeout -k 10 120 tools/ta_rate --footprint 6000000 > gpurun_out/r3p/ta_rate_6MB.txt 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --workload c4_miku --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3p/c4_64spp.json 2> gpurun_out/r3p/c4.err || exit 1
timeout -k 10 200 python3 bench.py --workload c5_composite --spp 16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3p/c5_16spp.json 2> gpurun_out/r3p/c5.err || exit 1
bash tools/pmc_workload.sh c3 "--workload c3_utah" > gpurun_out/r3p/pmc_c3.log 2>&1 || exit 1
bash tools/pmc_workload.sh c4 "--workload c4_miku --spp 64" > gpurun_out/r3p/pmc_c4.log 2>&1 || exit 1
bash tools/pmc_workload.sh c5 "--workload c5_composite --spp 8" > gpurun_out/r3p/pmc_c5.log 2>&1 || exit 1
echo done
