# usage: bash tools/gpu_encode_ab.sh <suffix>...   -> kernel time of the dominant encoder kernels per variant library (rocprofv3 stats)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sfx in "$@"; do
  O=$PWD/gpurun_out/enc_ab$sfx; rm -rf $O; mkdir -p $O
  L=trafficbots_amd/lib/libtrafficbots_hip$sfx.so; [ "$sfx" = "-" ] && L=trafficbots_amd/lib/libtrafficbots_hip.so
  (cd /tmp; TB_HIP_LIB=$GRAFT_REPO_ROOT/$L rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lean > $O/log.txt 2>&1)
  python - "$O" "$sfx" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][-40:]
    if 'polyline' in n or '_pl' in n: print(sys.argv[2], n, 'us/encode %.1f'%(float(r['TotalDurationNs'])/4e3))
PY
done
