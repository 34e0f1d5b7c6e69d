#!/bin/bash
# GPU box: the oracle-checked fuzzers of tests/probes under FULL guard -- every tensor handed to the C ABI and every carve of the
# library's workspace between unmapped pages, buffers at the end (1) and at the start (0) of their mappings.  Output: gpurun_out/guard_fuzz.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/guard_fuzz.txt; : > $O
for e in 1 0; do
  for job in "gpu_fuzz_shapes.py 48" "gpu_fuzz_validation.py 16" "gpu_fuzz_rules_post_metrics.py 12" "gpu_fuzz_warm_start.py 24" "gpu_fuzz_bf16.py"; do
    set -- $job
    echo "### at_end=$e $job" >> $O
    GUARD_AT_END=$e TB_WS_GUARD=$((2-e)) FUZZ_SEED=$((4200+e)) FUZZ_KEEP_GOING=1 GUARD_FUZZ_SCRIPT=$1 timeout 1500 python tests/probes/gpu_guard_fuzz.py $2 2>&1 \
      | grep -E "GUARD-FUZZ-OK|all [0-9]+ cases|^ok$|Memory access fault|Error|error|FAIL|OUTSIDE" | tail -6 >> $O
  done
done
cat $O
