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
# the step kernel's launches by kind (kernel trace): a rollout issues fused C(t)+A(t+1) launches, half launches (the C-only steps of the
# batched warm start, the last step) and one batched A launch; bench.py's kernel_us.k_step_fused is the average of the fused ones
for f in sorted(files):
    if not f.endswith("kernel_trace.csv"):
        continue
    dur = {}
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if kernel in name:
                dur.setdefault(name.split("(")[0], []).append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
    print(f"## {kernel}* launches by kind (us, kernel trace)")
    for name, v in dur.items():
        med = statistics.median(v)
        fused = [x for x in v if x > 0.75 * med and x < 1.5 * med] if len(v) > 20 else v
        rest = [x for x in v if not (x > 0.75 * med and x < 1.5 * med)] if len(v) > 20 else []
        print(f"{name[:60]:60s} n={len(v):5d}  fused-size launches: n={len(fused)} avg {sum(fused) / max(1, len(fused)):.2f}   "
              f"others: n={len(rest)} avg {sum(rest) / max(1, len(rest)):.2f}")
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
