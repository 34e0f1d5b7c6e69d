#!/bin/bash
# GPU box, round 6: weight-stationary hop micro-benchmark; the default-scheduler build (does profiles/r05_experiments.txt item 24 still
# fault?); the parity control (tools/parity_control.py) incl. the LayerNorm-association diagnosis build; every closed-loop golden on that
# build against the shipped one; the xfail-strict fuzz draws.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06
mkdir -p $O
(timeout 120 tools/microtests/bin/ws_hop 128 536 > $O/ws_hop.txt 2>&1; echo "rc=$?" >> $O/ws_hop.txt)
D=trafficbots_amd/lib/libtrafficbots_hip_defsched.so
(TB_HIP_LIB=$D timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "against_reference_golden and (headline_2 or small_k1 or headline_k6)" > $O/defsched_tests.txt 2>&1; echo "rc=$?" >> $O/defsched_tests.txt)
(TB_HIP_LIB=$D timeout 600 python bench.py --lean --steps 10 --warmup 3 --no-cpu-baseline > $O/defsched_bench.json 2> $O/defsched_bench.err; echo "rc=$?" >> $O/defsched_bench.err)
(TB_HIP_LIB=$D timeout 600 python bench.py --lean --steps 5 --warmup 2 --no-cpu-baseline --operand-precision bf16 > $O/defsched_bench_bf16.json 2>> $O/defsched_bench.err; echo "rc=$?" >> $O/defsched_bench.err)
(timeout 1800 python tools/parity_control.py --lib lnorder=trafficbots_amd/lib/libtrafficbots_hip_dbg_lnorder.so --json $O/parity_control.json > $O/parity_control.txt 2>&1; echo "rc=$?" >> $O/parity_control.txt)
for v in "" _dbg_lnorder; do
  L=""; [ -n "$v" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$v.so"
  env $L timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "against_reference_golden or suite_level or zz_write_report" > $O/goldens$v.log 2>&1
  cp gpurun_out/parity_report.json $O/parity_report$v.json
done
(timeout 1500 python -m pytest tests/test_gpu_configs.py -q -m gpu -p no:cacheprovider -k "known_outside or lone" -rxX > $O/xfail_tests.txt 2>&1; echo "rc=$?" >> $O/xfail_tests.txt)
tail -3 $O/ws_hop.txt $O/defsched_tests.txt $O/defsched_bench.err $O/xfail_tests.txt
