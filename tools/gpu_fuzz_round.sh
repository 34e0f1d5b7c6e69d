cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_r02.txt; : > $O
run() { echo "### $*" >> $O; ( eval "$@" ) 2>&1 | tail -4 >> $O; }
run FUZZ_SEED=5150 timeout 900 python tests/probes/gpu_fuzz_shapes.py 48
run FUZZ_SEED=8086 timeout 900 python tests/probes/gpu_fuzz_validation.py 24
run FUZZ_SEED=17 timeout 600 python tests/probes/gpu_fuzz_bf16.py
run FUZZ_SEED=6502 timeout 900 python tests/probes/gpu_fuzz_rules_post_metrics.py 16
run FUZZ_SEED=68000 timeout 900 python tests/probes/gpu_fuzz_warm_start.py 32
run STRADDLE_B=40 timeout 600 python tests/probes/gpu_straddle_check.py
cat $O
