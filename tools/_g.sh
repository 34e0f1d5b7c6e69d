for rep in 1 2; do
for lib in "" _base; do
  L=""; [ -n "$lib" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$lib.so"
  env $L python bench.py --only-config stress_bf16 --config-steps 5 --no-cpu-baseline --lean 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lib=$lib stress', r.get('value'), r.get('k_step_fused_us'))
"
done
done
