#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: compile exp/src_NAME/ (a copy of csrc with local edits) into exp/NAME.so
set -e
n=$1; shift
cd "$(dirname "$0")/../exp/src_$n"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -I. -shared -o ../$n.so rt_kernels.hip rt_bvh_device.hip -x hip rt_api.cpp rt_bvh.cpp 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|" | head -20
ls -la ../$n.so
