"""Dump the per-kernel statistics of a rocprofv3 rocpd database (--kernel-trace --stats) as text."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats  (source: {db}); durations in microseconds\n")
    f.write(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel\n")
    for name, calls, total, avg, pct in rows:
        if len(name) > 110:
            name = name[:107] + "..."
        f.write(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:7.2f}  {name}\n")
    f.write("\n# per-kernel resources (first dispatch of each)\n")
    seen = set()
    for r in con.execute("select name, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels"):
        if r[0] in seen or not r[0].startswith("tb::"):
            continue
        seen.add(r[0])
        f.write(f"{r[0][:60]:60s} grid=({r[1]},{r[2]}) wg={r[3]} lds={r[4]} vgpr={r[5]} agpr={r[6]} sgpr={r[7]}\n")
print(open(out).read())
