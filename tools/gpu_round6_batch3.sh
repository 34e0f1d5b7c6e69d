#!/bin/bash
# GPU box, round 6: the new golden + suite rule, the default-scheduler diagnosis builds, ws_hop, one full bench line.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "against_reference_golden or suite_level or zz_write_report" > $O/goldens_5.log 2>&1; echo "rc=$?" >> $O/goldens_5.log; cp gpurun_out/parity_report.json $O/parity_report_5.json)
for v in _defsched_wz _defsched_novf; do
  (TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "against_reference_golden and (headline_2 or small_k1)" > $O/defsched$v.txt 2>&1; echo "rc=$?" >> $O/defsched$v.txt)
done
(timeout 180 tools/microtests/bin/ws_hop 128 536 > $O/ws_hop.txt 2>&1; echo "rc=$?" >> $O/ws_hop.txt)
(timeout 1500 python bench.py > $O/bench_mid.json 2> $O/bench_mid.err; echo "rc=$?" >> $O/bench_mid.err)
(TB_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --configs > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "rc=$?" >> $O/bench_gloo2.err)
tail -n 3 $O/goldens_5.log $O/defsched_defsched_wz.txt $O/defsched_defsched_novf.txt $O/bench_mid.err $O/bench_gloo2.err
