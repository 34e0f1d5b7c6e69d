#!/bin/bash
# GPU box: VERDICT r04 task 4 (d) -- which non-GEMM arithmetic carries the distance of the HIP path to the reference's base fp32 run?
# Diagnosis builds of the library (tools/build_variant.sh _dbg_<name> -DTB_DBG_...): softmax with the library expf in natural units,
# the pose-PE sincos from the fp32 library instead of the fp64 polynomial, another association of the LayerNorm row sums -- one at a
# time -- run through the closed-loop goldens that sit at the edge of the rule; per build the rank / ratio of every case is printed.
# usage: bash tools/gpu_bias_probe.sh   -> gpurun_out/r05/bias_probe.txt
mkdir -p gpurun_out/r05
for v in "" _dbg_expf _dbg_sincosf _dbg_lnorder; do
  L=""; [ -n "$v" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$v.so"
  env $L timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
      -k "(against_reference_golden and (headline_8 or headline_2 or headline_k6 or small_k1 or headline_w_trained or masks_k3)) or zz_write_report" > gpurun_out/r05/bias_$v.log 2>&1
  cp gpurun_out/parity_report.json gpurun_out/r05/bias_report$v.json
  python - <<PY
import json
r = json.load(open("gpurun_out/r05/bias_report$v.json"))
print("== build '$v'")
for k in ("small_k1", "masks_k3", "headline_2", "headline_8", "headline_k6", "headline_w_trained"):
    c = r.get(k, {})
    print(f"  {k:20s} vs fp32 {c.get('final_vs_fp32', float('nan')):.3e} rank {c.get('rank_vs_fp32')} ratio {c.get('ratio_to_median_vs_fp32', float('nan')):.2f} | "
          f"vs fp64 {c.get('final_vs_fp64', float('nan')):.3e} rank {c.get('rank_vs_fp64')} ratio {c.get('ratio_to_median_vs_fp64', float('nan')):.2f}")
PY
  tail -1 gpurun_out/r05/bias_$v.log
done 2>&1 | tee gpurun_out/r05/bias_probe.txt
