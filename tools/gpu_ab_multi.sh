# usage: bash tools/gpu_ab_multi.sh <suffix>...  ("-" = the shipped library): headline bench per variant library, two repetitions
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  for prec in fp32 bf16; do
    L=""; [ "$lib" != "-" ] && L="TB_HIP_LIB=trafficbots_amd/lib/libtrafficbots_hip$lib.so"
    env $L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean --operand-precision $prec 2>gpurun_out/ab.err | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib $prec value %.0f launch_us %.2f'%(r['value'], r['roofline']['avg_launch_us']))"
  done
done
done
