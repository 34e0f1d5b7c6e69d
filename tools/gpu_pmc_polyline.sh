#!/bin/bash
# GPU box: rocprofv3 --pmc passes (counters only, separate passes) over the encoders of a short bench run, per library variant;
# prints the per-launch averages of the polyline kernel.  usage: bash tools/gpu_pmc_polyline.sh <suffix>...   ("-" = the shipped library)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sfx in "$@"; do
  L=trafficbots_amd/lib/libtrafficbots_hip$sfx.so; [ "$sfx" = "-" ] && L=trafficbots_amd/lib/libtrafficbots_hip.so
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    O=$PWD/gpurun_out/pmc_pl; rm -rf $O; mkdir -p $O
    (cd /tmp; TB_HIP_LIB=$GRAFT_REPO_ROOT/$L timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > $O/log.txt 2>&1)
    python - "$O" "$sfx" <<'PY'
import csv,glob,sys,collections
fs=glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True)
if not fs: print(sys.argv[2], 'no counter file'); sys.exit()
acc=collections.defaultdict(lambda: [0.0,0])
for r in csv.DictReader(open(fs[0])):
    if 'polyline' not in r['Kernel_Name']: continue
    k=r['Counter_Name']; acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
for k,(v,n) in sorted(acc.items()):
    # rows are per (dispatch, counter) or per (dispatch, counter, dimension): the sum over rows / dispatches
    print(f"{sys.argv[2]:10s} {k:28s} sum {v:.4e} rows {n}")
PY
  done
done
