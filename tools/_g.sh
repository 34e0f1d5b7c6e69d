mkdir -p gpurun_out/r05
timeout 300 python tests/probes/gpu_aw_check.py 30 2>&1 | grep -v amdgpu.ids
for aw in 1 0 1 0; do
  TB_STEP_AW=$aw python bench.py --only-config stress_bf16 --config-steps 5 --no-cpu-baseline --lean 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stress AW=$aw', r.get('value'), r.get('ms_per_pass'))
"
done
