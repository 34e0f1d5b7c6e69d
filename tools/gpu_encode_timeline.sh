# GPU box: kernel timeline of the last tb_encode_scene of a short bench run (rocprofv3 --kernel-trace): start / end / duration per launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$PWD/gpurun_out/enc_tl; rm -rf $O; mkdir -p $O
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > $O/log.txt 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/enc_tl/**/*kernel_trace.csv', recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-34:], r.get('Queue_Id','?')) for r in csv.DictReader(open(f))]
rows.sort()
idx=[i for i,r in enumerate(rows) if 'polyline_fused' in r[2]]
i0=idx[-1]
if len(idx) > 1 and rows[idx[-1]][0] - rows[idx[-2]][1] < 50000: i0 = idx[-2]  # (two halves of one encode)
j=i0
for back in range(1,8):
    if 'k_encode_tokens' in rows[i0-back][2]: j=i0-back
t0=rows[j][0]
for r in rows[j:j+48]:
    if 'k_step' in r[2] or 'rollout' in r[2] or 'at::' in r[2]: break  # (the encode is over: rollout prologue / torch kernels)
    print(f"{(r[0]-t0)/1e3:8.1f} -> {(r[1]-t0)/1e3:8.1f}  dur {(r[1]-r[0])/1e3:7.1f}  q{r[3]} {r[2]}")
PY
