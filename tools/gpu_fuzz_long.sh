cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_r02b.txt; : > $O
run() { echo "### $*" >> $O; ( eval "$@" ) 2>&1 | tail -2 >> $O; }
run FUZZ_SEED=424242 timeout 1700 python tests/probes/gpu_fuzz_shapes.py 160
run FUZZ_SEED=7 timeout 1700 python tests/probes/gpu_fuzz_validation.py 48
run FUZZ_SEED=1234 timeout 900 python tests/probes/gpu_fuzz_rules_post_metrics.py 32
run FUZZ_SEED=55 timeout 900 python tests/probes/gpu_fuzz_warm_start.py 96
run FUZZ_SEED=3 timeout 600 python tests/probes/gpu_fuzz_bf16.py
cat $O
