#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, the bench line (with CPU baseline), then rocprofv3 on a short bench run:
# one --kernel-trace --stats pass and three separate --pmc passes (never combined with other traces).  Output: gpurun_out/
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/tests_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/tests_gpu.log)
tail -n 3 gpurun_out/tests_gpu.log
(timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?")
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
OUT=$PWD/gpurun_out/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
cd - > /dev/null
find gpurun_out/prof -name "*.csv" | head -20
python tools/prof_pmc_summary.py gpurun_out/prof > gpurun_out/prof/summary.txt 2>&1
cat gpurun_out/prof/summary.txt | head -60
