#!/bin/bash
# usage: tools/build_variant.sh <suffix> [extra -D flags]   -> trafficbots_amd/lib/libtrafficbots_hip<suffix>.so  (same sources / flags as __graft_entry__.build)
set -e
cd "$(dirname "$0")/../trafficbots_amd/csrc"
sfx="$1"; shift
SRC=$(python3 -c "import sys; sys.path.insert(0, '../..'); import __graft_entry__ as g; print(' '.join(g.SOURCES))")
# flags: toolchain.json (TB_DROP_FLAGS="-amdgpu-sched-strategy=max-ilp" etc. remove entries for A/B builds)
FLAGS=$(python3 -c "import sys; sys.path.insert(0, '../..'); import __graft_entry__ as g; print(' '.join(g.FLAGS))")
/opt/rocm/bin/hipcc $FLAGS -shared "$@" -o ../lib/libtrafficbots_hip${sfx}.so $SRC
