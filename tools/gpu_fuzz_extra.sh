cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_${FUZZ_ROUND:-r05}_extra.txt; : > $O
run() { echo "### $*" >> $O; ( eval "$@" ) 2>&1 | grep -E "OUTSIDE|FAIL|all [0-9]+ cases|worst|Error|error|^ok$" | tail -12 >> $O; }
run FUZZ_SEED=101 FUZZ_KEEP_GOING=1 timeout 1500 python tests/probes/gpu_fuzz_validation.py 100
run FUZZ_SEED=102 FUZZ_KEEP_GOING=1 timeout 1500 python tests/probes/gpu_fuzz_validation.py 100
run FUZZ_SEED=103 timeout 1500 python tests/probes/gpu_fuzz_shapes.py 200
run FUZZ_SEED=104 timeout 900 python tests/probes/gpu_fuzz_warm_start.py 100
run FUZZ_SEED=105 timeout 900 python tests/probes/gpu_fuzz_rules_post_metrics.py 48
run FUZZ_SEED=106 timeout 600 python tests/probes/gpu_fuzz_bf16.py
cat $O
