#!/bin/bash
# usage: ab_env.sh WORKLOAD SPP "ENV=.. ENV=.." ...
wl=$1; spp=$2; shift 2
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 bench.py --workload $wl --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --no-count-pass 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['stage_ms_per_step_rank0'])" || exit 1
done
