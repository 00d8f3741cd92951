#!/bin/bash
# pmc_workload.sh TAG "BENCH ARGS": rocprofv3 kernel trace and hardware-counter passes over one short bench run each
# (run on the GPU box).  Every pass is a run of its own and stays within what one run can collect (8 SQ, 4 TCC, 4 TCP,
# 2 TA, 2 GRBM counters).  Results: gpurun_out/pmc_TAG/{trace,a,b,c,d,e}/...csv; summarise with tools/pmc_report.py.
set -o pipefail
tag=$1; args=$2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bench="python3 $GRAFT_REPO_ROOT/bench.py $args --steps 1 --warmup 0 --no-cpu-baseline --no-count-pass"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- $bench > $out/trace.json 2> $out/trace.err || exit 1
pass() { # NAME COUNTERS...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- $bench > $out/$name.json 2> $out/$name.err || exit 1
}
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
pass b SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE GRBM_TA_BUSY
pass c TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass d TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass e FETCH_SIZE SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVES
pass f WRITE_SIZE SQ_WAVES
ls $out
