#!/bin/bash
# usage: tools/build_variant.sh <suffix> [extra -D flags]   -> trafficbots_amd/lib/libtrafficbots_hip<suffix>.so
set -e
cd "$(dirname "$0")/../trafficbots_amd/csrc"
sfx="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared "$@" -o ../lib/libtrafficbots_hip${sfx}.so tb_api.hip tb_rollout_kernels.hip tb_stepx_kernels.hip tb_stepx_bf16_kernels.hip tb_rules_kernels.hip tb_post_kernels.hip tb_metrics_kernels.hip tb_train_kernels.hip tb_encode_kernels.hip tb_encodex_kernels.hip
