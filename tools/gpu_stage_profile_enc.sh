#!/bin/bash
# GPU box: stage profile of k_polyline_fused8<true> -- clock64 stamps of thread 0 of workgroup 700, printed from the device by the
# -DTB_PROFILE_ENC build (tools/build_variant.sh _encprof -DTB_PROFILE_ENC); headline shape, 4096 workgroups on 256 CUs.
cd $GRAFT_REPO_ROOT
TB_HIP_LIB=$PWD/trafficbots_amd/lib/libtrafficbots_hip_encprof.so timeout 300 python bench.py --steps 2 --warmup 1 --lean --no-cpu-baseline 2>&1 | grep ENCPROF8 | head -${1:-8}
