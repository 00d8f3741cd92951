#!/bin/bash
# cpu_baselines.sh: the oracle's throughput (16 threads and 1 thread) on a bounded sample of every workload of the bench
# table, beside the GPU figure of the same run (run on the GPU box)
cd $GRAFT_REPO_ROOT
for spec in "c1_cornell 0" "c2_whitted 0" "c3_utah 0" "c4_miku 64" "c5_composite 8"; do
  set -- $spec
  extra=""; [ "$2" != "0" ] && extra="--spp $2"
  timeout -k 10 280 python3 bench.py --workload $1 $extra --steps 1 --warmup 0 --cpu-seconds 12 --no-count-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; b=d['cpu_baseline']
print(c['workload'], '| GPU', d['value'], 'Mrays/s | CPU', b['cores'], 'threads', b['value'], 'Mrays/s | CPU 1 thread', b['single_thread']['value'], 'Mrays/s | pruned', c['zero_weight_rays_pruned']['fraction'], '|', b['sample'][:70])" || exit 1
done
