"""Summarise a directory of rocprofv3 CSV outputs (one --kernel-trace --stats pass, several --pmc passes) as text:
per-kernel statistics and, per counter, the median / max over the launches of the step kernel."""
import csv
import glob
import os
import statistics
import sys

root = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_step"
files = glob.glob(os.path.join(root, "**", "*.csv"), recursive=True)
for f in sorted(files):
    if f.endswith("kernel_stats.csv"):
        print("## kernel stats (ns)")
        for i, line in enumerate(open(f)):
            if i < 14:
                print(line.rstrip()[:190])
        print()
counters = {}
for f in sorted(files):
    if "counter_collection" not in f:
        continue
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        for row in rd:
            name = row.get("Kernel_Name", "")
            if kernel not in name:
                continue
            c, v = row.get("Counter_Name"), row.get("Counter_Value")
            if c is None or v is None:
                continue
            counters.setdefault(c, []).append(float(v))
print(f"## PMC, {kernel}*, per launch")
for c in sorted(counters):
    v = counters[c]
    print(f"{c:28s} n={len(v):5d}  median {statistics.median(v):.6g}   max {max(v):.6g}")
