for w in 1 0 1 0; do
  TB_STEP_WARM=$w python bench.py --steps 30 --warmup 5 --no-cpu-baseline --configs 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WARM=$w headline', r['value'], r['roofline']['avg_launch_us'], 'two streams', r['two_batches_in_flight']['value'])
"
done
