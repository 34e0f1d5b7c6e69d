cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$PWD/gpurun_out/enc_stats; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lean > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/enc_stats/**/*kernel_stats.csv', recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][-44:]
    if 'k_step' in n or 'at::' in r['Name'] or 'rocclr' in n: continue
    t=float(r['TotalDurationNs'])/1e3/4; tot+=t
    if t>3: print(f"{n:46s} calls {r['Calls']:>4s} us/encode {t:8.1f}")
print('sum', tot)
PY
