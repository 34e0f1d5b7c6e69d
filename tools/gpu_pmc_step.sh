#!/bin/bash
# GPU box (via gpurun): rocprofv3 passes over a short bench run of the FINAL build -- one --kernel-trace --stats pass and separate
# --pmc passes (never combined with other traces) -- for the fp32-accurate and the bf16 step kernel.
# Output: gpurun_out/pmc_step/{fp32,bf16}/summary.txt and gpurun_out/pmc_step/pmc_step_kernel.json (copied to profiles/ by the builder;
# bench.py reads profiles/pmc_step_kernel.json for `roofline.traffic` and `roofline.mfma_busy.measured`).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for prec in fp32 bf16; do
  BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lean --operand-precision $prec"
  O=$OUT/$prec; mkdir -p $O
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > $O/pmc_write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- $BENCH > $O/pmc_sq.log 2>&1
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $BENCH > $O/pmc_tcc.log 2>&1
  timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc_tcp -- $BENCH > $O/pmc_tcp.log 2>&1
  cd $ROOT
  python tools/prof_pmc_summary.py $O > $O/summary.txt 2>&1
done
python tools/prof_pmc_json.py $OUT > $OUT/pmc_step_kernel.json
cat $OUT/fp32/summary.txt | head -40
cat $OUT/pmc_step_kernel.json
