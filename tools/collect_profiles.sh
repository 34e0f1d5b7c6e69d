#!/bin/bash
# usage: tools/collect_profiles.sh rNN   -- copies what tools/gpu_pmc_step.sh merged into gpurun_out/pmc_step/ to profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
r=$1
mkdir -p profiles/${r}_pmc
for n in fp32 bf16 k6_fp32 k6_bf16 stress_fp32 stress_bf16; do
  [ -f gpurun_out/pmc_step/$n/summary.txt ] && cp gpurun_out/pmc_step/$n/summary.txt profiles/${r}_pmc/${n}_summary.txt
done
cp gpurun_out/pmc_step/pmc_step_kernel.json profiles/pmc_step_kernel.json
for n in fp32 bf16; do
  f=$(ls -t $(find gpurun_out/pmc_step/$n/stats -name "*kernel_stats.csv") | head -1)  # (the newest: earlier calls leave theirs behind)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --lean --operand-precision $n  (tools/gpu_pmc_step.sh, round ${r})"; cat $f; } > profiles/${r}_rocprof_$n.txt
done
ls -la profiles/${r}_pmc profiles/${r}_rocprof_*.txt
