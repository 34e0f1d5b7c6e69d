#!/bin/bash
# usage: tools/isa_stats.sh <file.hip> <mangled-kernel-prefix> [extra flags]  -> instruction histogram + register usage of one kernel
set -e
src=$1; sym=$2; shift 2
cd /root/repo/trafficbots_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-ilp -S --cuda-device-only "$@" $src -o /tmp/isa_stats.s 2>/dev/null
awk -v s="^$sym" '$0 ~ s":"{f=1} /^\.Lfunc_end/{if(f){f=0}} f{print}' /tmp/isa_stats.s > /tmp/isa_kernel.s
echo "instructions: $(grep -v '^\s*;' /tmp/isa_kernel.s | awk '{print $1}' | grep -v '^\.\|^_Z' | grep -c .)   mfma: $(grep -c v_mfma /tmp/isa_kernel.s)"
grep -v '^\s*;' /tmp/isa_kernel.s | awk '{print $1}' | grep -v '^\.\|^_Z' | sed 's/_e32$//; s/_e64$//' | sort | uniq -c | sort -rn | head -${TOP:-24}
grep -A14 "^$sym:" /tmp/isa_stats.s >/dev/null; grep -E "\.(sgpr|vgpr|agpr)_count|spill_count|NumVgprs|NumAgprs|ScratchSize" /tmp/isa_stats.s | head -12
