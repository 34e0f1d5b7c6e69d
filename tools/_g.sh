timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "warm_start or carve or graph_replay or helper_workgroups or golden or config3" 2>&1 | tail -6
for pi in 1 0 1 0; do
  TB_STEP_PRE_INTER=$pi python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PRE_INTER=$pi headline', r['value'], r['ms_per_step'])
"
done
for pi in 1 0; do
  TB_STEP_PRE_INTER=$pi python bench.py --only-config k6_bf16 --config-steps 5 --no-cpu-baseline --lean 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PRE_INTER=$pi k6_bf16', r.get('value'), r.get('ms_per_pass'))
"
done
