#!/bin/bash
# GPU box, end of round 6: a longer soak and more seeds of the round-6 fuzz probes (evidence beyond tools/gpu_round6_final.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r06/extra_fuzz.txt
mkdir -p gpurun_out/r06; : > $O
run() { echo "### $*" >> $O; ( eval "$@" ) 2>&1 | grep -E "soak:|growth|memory|cases, [0-9]+ failed|FAIL|OUTSIDE|ensemble:|Error|Traceback|no growth|MISMATCH" | tail -30 >> $O; }
run N=12000 timeout 1500 python tests/probes/gpu_soak.py
run REPS=40 timeout 900 python tests/probes/gpu_soak2.py
for s in 780 781 782 783 784; do run FUZZ_SEED=$s timeout 900 python tests/probes/gpu_fuzz_cases.py 150; done
for s in 992 993; do run FUZZ_SEED=$s timeout 900 python tests/probes/gpu_fuzz_training.py 100; done
run FUZZ_SEED=314 timeout 900 python tests/probes/gpu_fuzz_stepwise.py 100
run FUZZ_CUR=5 FUZZ_SEED=22 FUZZ_KEEP_GOING=1 timeout 1500 python tests/probes/gpu_fuzz_validation.py 40
cat $O
