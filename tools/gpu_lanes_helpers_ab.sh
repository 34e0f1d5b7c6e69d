# A/B of the helper workgroups under lanes (profiles/r06_experiments.txt item 14): TB_STEP_HELPERS unset = automatic (off while a neighbour
# context is launching), 1 = on regardless, 0 = off
mkdir -p gpurun_out/r06
O=gpurun_out/r06/lanes_helpers_ab.txt
: > $O
for rep in 1 2; do
for L in 2 3; do
  for H in "" 1 0; do
    for K in 1; do
      if [ -z "$H" ]; then env K=$K LANES=$L N=24 REPS=2 timeout 300 python tests/probes/gpu_e2e_prefetch_loop.py 2>/dev/null | tail -1 >> $O
      else env TB_STEP_HELPERS=$H K=$K LANES=$L N=24 REPS=2 timeout 300 python tests/probes/gpu_e2e_prefetch_loop.py 2>/dev/null | tail -1 >> $O; fi
    done
  done
done
done
cat $O
