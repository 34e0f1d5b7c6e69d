for lib in "" _base; do
  for prec in fp32 bf16; do
    L=""; [ -n "$lib" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$lib.so"
    env $L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --operand-precision $prec --configs stress_$prec --config-steps 3 > gpurun_out/ab_${prec}${lib}.json 2>gpurun_out/ab.err
    python - <<PY
import json
r=json.loads(open("gpurun_out/ab_${prec}${lib}.json").read().strip().splitlines()[-1])
c=r.get("configs",{})
print("lib='$lib' $prec value %.0f kernel_us %s enc_ms %.3f"%(r["value"], r["roofline"]["avg_launch_us"], r["encode_roofline"]["encode_ms"]), {k:(v.get("value"),v.get("kernel_us")) for k,v in c.items()})
PY
  done
done
