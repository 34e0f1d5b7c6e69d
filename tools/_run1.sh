cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for lib in "" _ldp144; do
  L="A=1"; [ -n "$lib" ] && L="TB_HIP_LIB=$PWD/trafficbots_amd/lib/libtrafficbots_hip$lib.so"
  env $L python tests/probes/gpu_encode_time.py 2>&1 | tail -2 | sed "s/^/lib='$lib' /"
  for prec in fp32 bf16; do
    env $L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --operand-precision $prec > gpurun_out/ab_${prec}${lib}.json 2>gpurun_out/ab.err
    python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/ab_${prec}${lib}.json").read().strip().splitlines()[-1])
    print("lib='$lib' $prec value %.0f launch_us %.2f encode_ms %.3f"%(r["value"], r["roofline"]["avg_launch_us"], r["encode_ms"]))
except Exception as e:
    print("lib='$lib' $prec FAILED", e); print(open("gpurun_out/ab.err").read()[-600:])
PY
  done
done
done
