#!/bin/bash
# knob_ab.sh "NAME=VALUE ..." ["NAME=VALUE ..." ...]: the three mesh workloads under each set of environment knobs
# (run on the GPU box); prints time per frame and the walk's work per ray from the work-counting pass
cd $GRAFT_REPO_ROOT
run() { python3 bench.py "$@" --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], 'ms', d['value'], d['unit'], 'nodes/ray', r.get('nodes_per_ray'), 'tris/ray', r.get('tris_per_ray'))"; }
for set in "$@"; do
  echo "== $set"
  (
    for kv in $set; do export "$kv"; done
    echo -n "c4_miku 64 spp: "; run --workload c4_miku --spp 64 --steps 3 || exit 1
    echo -n "c5_composite 16 spp: "; run --workload c5_composite --spp 16 --steps 3 || exit 1
    echo -n "c3_utah 256 spp: "; run --steps 3 || exit 1
  ) || exit 1
done
