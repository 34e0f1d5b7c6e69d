mkdir -p gpurun_out/r06
O=gpurun_out/r06/lanes_helpers_ab.txt
: > $O
for L in 2 3 4; do
  for H in "" 0; do
    for K in 1 6; do
      if [ -z "$H" ]; then env K=$K LANES=$L N=24 REPS=2 timeout 300 python tests/probes/gpu_e2e_prefetch_loop.py 2>/dev/null | tail -1 >> $O
      else env TB_STEP_HELPERS=$H K=$K LANES=$L N=24 REPS=2 timeout 300 python tests/probes/gpu_e2e_prefetch_loop.py 2>/dev/null | tail -1 >> $O; fi
    done
  done
done
cat $O
