#!/bin/bash
# GPU box (via gpurun): rocprofv3 passes over short bench runs of the CURRENT build -- one --kernel-trace --stats pass and separate
# --pmc passes (never combined with other traces) -- for the step kernel at the headline shape (fp32-accurate and bf16 operands) and
# at the BASELINE configs[3] / configs[4] shapes (K=6 and stress, both precisions).
# Output: gpurun_out/pmc_step/<name>/summary.txt and gpurun_out/pmc_step/pmc_step_kernel.json, stamped with the SHA-256 of the library
# it was collected on (copied to profiles/ by the builder; bench.py uses it for `roofline.traffic`, `mfma_busy.measured` and the
# `configs[*].roofline.traffic` fields only when that hash is the hash of the library it is timing: `pmc_matches_build`).
# Usage: tools/gpu_pmc_step.sh [names...]   (default: all six)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
NAMES=${@:-fp32 bf16 k6_fp32 k6_bf16 stress_fp32 stress_bf16}
for name in $NAMES; do
  case $name in
    fp32|bf16) BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lean --operand-precision $name"; FULL=1 ;;
    *)         BENCH="python $ROOT/bench.py --only-config $name --config-steps 2 --no-cpu-baseline --lean"; FULL=0 ;;
  esac
  O=$OUT/$name; mkdir -p $O
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > $O/pmc_write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- $BENCH > $O/pmc_sq.log 2>&1
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $BENCH > $O/pmc_tcc.log 2>&1
  if [ $FULL = 1 ]; then
    timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc_tcp -- $BENCH > $O/pmc_tcp.log 2>&1
  fi
  cd $ROOT
  python tools/prof_pmc_summary.py $O > $O/summary.txt 2>&1
done
python tools/prof_pmc_json.py $OUT > $OUT/pmc_step_kernel.json
cat $OUT/fp32/summary.txt 2>/dev/null | head -40
cat $OUT/pmc_step_kernel.json
