timeout 300 python tests/probes/gpu_aw_check.py 30 2>&1 | grep -v amdgpu.ids | tail -2
for c in stress_bf16 k6_bf16 stress_bf16 k6_bf16; do
  python bench.py --only-config $c --config-steps 5 --no-cpu-baseline --lean 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$c', r.get('value'), r.get('ms_per_pass'), r.get('k_step_fused_us'))
"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --operand-precision bf16 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline bf16', r.get('value'), r['roofline']['avg_launch_us'])
"
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "bf16 or carve or stress or config3 or config4 or warm_start or k6 or golden" 2>&1 | tail -6
