#!/bin/bash
# GPU box, end of round 6, on the FINAL build: bench line (driver's command), rocprofv3 kernel trace + PMC passes (tools/gpu_pmc_step.sh),
# kernel trace of the end-to-end loops, stage profile of the fused step launch, fuzz campaign.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD
O=gpurun_out/r06
mkdir -p $O
(timeout 1500 python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "rc=$?" >> $O/bench_final.err)
(TB_STAGE_JSON=$O/stage_constants.json TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip_prof.so timeout 600 python tools/gpu_stage_profile.py > $O/stage_profile_k_step_x.txt 2>&1; echo "rc=$?" >> $O/stage_profile_k_step_x.txt)
export TMPDIR=/tmp
for mode in plain lanes; do
  cd /tmp
  if [ $mode = plain ]; then E="PREFETCH=0 REPS=1"; else E="LANES=2 REPS=1"; fi
  env $E timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/prof_e2e_$mode -o e2e -- python $ROOT/tests/probes/gpu_e2e_prefetch_loop.py > $ROOT/$O/prof_e2e_$mode.log 2>&1
  cd $ROOT
done
bash tools/gpu_pmc_step.sh > $O/pmc_step.log 2>&1
(FUZZ_ROUND=r06 timeout 4000 bash tools/gpu_fuzz_long.sh > $O/fuzz.log 2>&1; echo "rc=$?" >> $O/fuzz.log)
(timeout 2400 python bench.py > $O/bench_final2.json 2> $O/bench_final2.err; echo "rc=$?" >> $O/bench_final2.err)
tail -n 3 $O/bench_final.err $O/stage_profile_k_step_x.txt $O/fuzz.log $O/bench_final2.err
ls $O/prof_e2e_plain $O/prof_e2e_lanes 2>/dev/null | head
