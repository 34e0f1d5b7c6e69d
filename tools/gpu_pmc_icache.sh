cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_PREFETCH|IB_|INSTS_|WAIT)[A-Z_0-9]*)" | sort -u > $O/avail.txt
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --lean"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
cd $R
python - <<'PY'
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob(f"gpurun_out/icache/{p}/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQC_ICACHE_REQ","SQ_IFETCH"): n[k]+=1
        for k in acc:
            if "k_step_x" in k: print(p,k,n[k],{c:v/max(n[k],1) for c,v in acc[k].items()})
PY
cat gpurun_out/icache/avail.txt | tr '\n' ' '
