#!/bin/bash
# usage: tools/build_variant.sh <suffix> [extra -D flags]   -> trafficbots_amd/lib/libtrafficbots_hip<suffix>.so  (same sources / flags as __graft_entry__.build)
set -e
cd "$(dirname "$0")/../trafficbots_amd/csrc"
sfx="$1"; shift
SRC=$(python3 -c "import sys; sys.path.insert(0, '../..'); import __graft_entry__ as g; print(' '.join(g.SOURCES))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared "$@" -o ../lib/libtrafficbots_hip${sfx}.so $SRC
