cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 2 1 2 1 2; do echo "DEST_SIDE=$v"; TB_ENCODE_DEST_SIDE=$v python tests/probes/gpu_encode_time.py 2>&1 | tail -2; done
TB_ENCODE_DEST_SIDE=2 bash tools/gpu_encode_timeline.sh 2>&1 | tail -30
for v in 1 2; do TB_ENCODE_DEST_SIDE=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEST_SIDE=$v', d['value'], d['encode_ms'], d.get('two_batches_in_flight',{}).get('value'))"; done
