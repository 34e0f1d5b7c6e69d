#!/bin/bash
# GPU box (via gpurun): the oracle-checked fuzzers of tests/probes on the current build, summary in gpurun_out/fuzz_r03.txt
# (copied to profiles/r03_fuzz_summary.txt by the builder).  Closed-loop cases beyond 1e-4 m are judged against an ensemble of the
# oracle on re-ordered batches measured on the spot (tools/ensemble.py), as the test suite does.
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_${FUZZ_ROUND:-r05}.txt; : > $O
run() { echo "### $*" >> $O; ( eval "$@" ) 2>&1 | grep -E "ensemble:|OUTSIDE|FAIL|all [0-9]+ cases|[0-9]+ cases, [0-9]+ failed|worst|Error|error|^ok$|first 11 steps [0-9.e-]+ collided" | tail -40 >> $O; }
run FUZZ_SEED=424242 timeout 1500 python tests/probes/gpu_fuzz_shapes.py 120
run FUZZ_SEED=11 FUZZ_KEEP_GOING=1 timeout 2400 python tests/probes/gpu_fuzz_validation.py 60
run FUZZ_SEED=1234 timeout 900 python tests/probes/gpu_fuzz_rules_post_metrics.py 32
run FUZZ_SEED=55 timeout 900 python tests/probes/gpu_fuzz_warm_start.py 64
run FUZZ_SEED=3 timeout 600 python tests/probes/gpu_fuzz_bf16.py
# a second set of draws
run FUZZ_SEED=20260929 timeout 1500 python tests/probes/gpu_fuzz_shapes.py 120
run FUZZ_SEED=7 FUZZ_KEEP_GOING=1 timeout 1800 python tests/probes/gpu_fuzz_validation.py 40
run FUZZ_SEED=56 timeout 900 python tests/probes/gpu_fuzz_warm_start.py 64
# round 6: the case generator the oracle is held to the reference with (tools/fuzz_cases.py): edge scenes, config overrides, weight modes
run FUZZ_SEED=777 timeout 900 python tests/probes/gpu_fuzz_cases.py 120
run FUZZ_SEED=778 timeout 900 python tests/probes/gpu_fuzz_cases.py 120
run FUZZ_SEED=991 timeout 900 python tests/probes/gpu_fuzz_training.py 60
run FUZZ_SEED=313 timeout 900 python tests/probes/gpu_fuzz_stepwise.py 60
cat $O
