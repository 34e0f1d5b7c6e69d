#!/bin/bash
# usage: tools/isa_prologue.sh [extra -D flags]  -> vmcnt waits / scalar loads in the launch prologue of tb::xh::k_step_x<false,false>
cd /root/repo/trafficbots_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-ilp -S --cuda-device-only "$@" tb_stepx_kernels.hip -o /tmp/stepx2.s 2>&1 | grep -v "warning\|^$"
awk '/^_ZN2tb2x[a-z0-9]*8k_step_xILb0ELb0EEEvNS_8RolloutPEiii:/{f=1} /^\.Lfunc_end0/{f=0} f' /tmp/stepx2.s > /tmp/k00n.s
grep -n "s_waitcnt vmcnt\|sched_barrier\|s_barrier" /tmp/k00n.s | head -${N:-14}
grep -E "\.(sgpr|vgpr|agpr)_count|spill_count|scratch" /tmp/stepx2.s | head -8
