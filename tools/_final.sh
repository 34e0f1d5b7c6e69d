cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?")
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","encode_ms","encode_ms_back_to_back","pmc_matches_build")})
r=d["roofline"]; print({k:r.get(k) for k in ("achieved","frac","avg_launch_us","traffic","floor_us","bound")}, r.get("mfma_busy",{}).get("measured"))
print(d.get("cpu_baseline")); print(d.get("max_abs_traj_err",{}) if not isinstance(d.get("max_abs_traj_err"),dict) else {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in list(v.items())[:8]}) for k,v in d["max_abs_traj_err"].items()})
for k,v in d.get("configs",{}).items(): print(k, {kk:v.get(kk) for kk in ("value","k_step_fused_us","encode_ms")}, (v.get("roofline") or {}).get("traffic_over_algorithmic"), (v.get("roofline") or {}).get("mfma_busy_measured"))
print(d.get("two_batches_in_flight"))
PY
