#!/bin/bash
# other_workloads.sh: one bench line per workload of the bench table besides the headline one (run on the GPU box)
cd $GRAFT_REPO_ROOT
for spec in "c1_cornell 0" "c2_whitted 0" "head 0" "c4_miku 0" "c5_composite 64"; do
  set -- $spec
  extra=""; [ "$2" != "0" ] && extra="--spp $2"
  timeout -k 10 280 python3 bench.py --workload $1 $extra --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print(c['workload'], '|', d['ms_per_step'], 'ms |', d['value'], 'Mrays/s | stages', c['stage_ms_per_step_rank0'], '| nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'], '| cold', c['cold_time_to_image_ms'], 'ms | scene', c['scene_upload_ms'], 'ms | bytes', c['device_bytes'])" || exit 1
done
