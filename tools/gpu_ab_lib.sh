#!/bin/bash
# GPU box: A/B of two builds of the library in the same call -- trafficbots_amd/lib/libtrafficbots_hip.so against
# libtrafficbots_hip_base.so (a copy of the previous build) -- headline shape, fp32-accurate and bf16, fused-launch time from HIP events.
# usage: bash tools/gpu_ab_lib.sh [reps]
for rep in $(seq 1 ${1:-2}); do
for lib in "" _base; do
  for prec in fp32 bf16; do
    L=""; [ -n "$lib" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$lib.so"
    env $L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --operand-precision $prec > gpurun_out/ab_${prec}${lib}.json 2>gpurun_out/ab.err
    python - <<PY
import json
r=json.loads(open("gpurun_out/ab_${prec}${lib}.json").read().strip().splitlines()[-1])
print("lib='$lib' $prec value %.0f launch_us %.2f"%(r["value"], r["roofline"]["avg_launch_us"]))
PY
  done
done
done
