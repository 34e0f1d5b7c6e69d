#!/bin/bash
# GPU box: A/B of libtrafficbots_hip.so against libtrafficbots_hip_base.so on the headline and the sub-record shapes, back to back.
# usage: [LIBS="_base _mid"] bash tools/gpu_ab_cfg.sh [reps] [configs...]
reps=${1:-2}; shift
cfgs=${@:-"stress_bf16 k6_bf16"}
for rep in $(seq 1 $reps); do
for lib in "" ${LIBS:-_base}; do
  L=""; [ -n "$lib" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$lib.so"
  env $L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --lean 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=\'$lib\' headline %.0f launch_us %.2f' % (r['value'], r['roofline']['avg_launch_us']))"
  for c in $cfgs; do
    env $L python bench.py --only-config $c --config-steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=\'$lib\' $c %.0f launch_us %.2f' % (r['value'], r.get('k_step_fused_us', 0)))"
  done
done
done
