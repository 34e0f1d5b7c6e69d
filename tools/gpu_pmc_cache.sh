#!/bin/bash
# GPU box: L2 (TCC) and L1 (TCP) counters of the fused step kernel over a short bench run; separate --pmc passes, no other traces.
# Output: gpurun_out/pmc_cache/summary.txt
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/pmc_cache
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*sum\|TCP_[A-Z0-9_]*sum" | sort -u > $OUT/avail.txt
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $set | tr ' ' '+')
  timeout 500 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- $BENCH > $OUT/$tag.log 2>&1
done
cd - > /dev/null
python - <<'PY' > $OUT/summary.txt 2>&1
import csv, glob, statistics, collections
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_cache/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step_x<false, false>" in r.get("Kernel_Name", ""):
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(f"{k:34s} n={len(v):4d} median {statistics.median(v):14.0f} max {max(v):14.0f}")
PY
cat $OUT/summary.txt; wc -l $OUT/avail.txt; tail -3 $OUT/*.log | head -40
