#!/usr/bin/env python3
"""Copies what tools/gpu_round6_final.sh left under gpurun_out/ (scratch) into profiles/ (tracked): the bench line, the PMC / kernel
trace summaries (tools/collect_profiles.sh), the stage profile + its machine-readable twin, the kernel statistics of the end-to-end
loops, the end-to-end block of the bench line as a record of its own, the fuzz log.  Usage: python tools/collect_round6.py"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r06")
P = os.path.join(ROOT, "profiles")


def bench_line(path):
    lines = [x for x in open(path) if x.startswith("{")]
    return json.loads(lines[-1]), lines[-1]


def kernel_table(csv_path, top=28):
    rows = list(csv.DictReader(open(csv_path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["  calls     total_us     avg_us     pct  kernel"]
    for r in rows[:top]:
        t = float(r["TotalDurationNs"])
        out.append("%7d %12.1f %10.2f %7.2f  %s" % (int(r["Calls"]), t / 1e3, t / 1e3 / int(r["Calls"]), 100 * t / tot, r["Name"][:150]))
    lib = sum(float(r["TotalDurationNs"]) for r in rows if "tb::" in r["Name"])
    out.append("  %d distinct kernels, %.1f us in total; library kernels (tb::) %.1f %% of the kernel time; %d calls of non-library kernels"
               % (len(rows), tot / 1e3, 100 * lib / tot, sum(int(r["Calls"]) for r in rows if "tb::" not in r["Name"])))
    return "\n".join(out)


def main():
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "collect_profiles.sh"), "r06"])
    # ---- the bench line: the SECOND run of the batch
    src = os.path.join(G, "bench_final2.json")
    d, raw = bench_line(src)
    open(os.path.join(P, "r06_bench_final.json"), "w").write(raw)
    # ---- stage profile
    # profiles/stage_constants.json is an INPUT of the build (tools/gen_warm_schedule.py -> csrc/tb_warm_schedule.inc, part of the source
    # fingerprint): adopting a new stage profile changes the sources the PMC / bench evidence of this very batch was taken on.  The
    # profile of the final sources is therefore kept beside it; `--adopt-stage-profile` replaces the build input (then regenerate the
    # schedule, rebuild, and collect again).
    shutil.copy(os.path.join(G, "stage_constants.json"), os.path.join(P, "r06_stage_constants_final.json"))
    if "--adopt-stage-profile" in sys.argv:
        shutil.copy(os.path.join(G, "stage_constants.json"), os.path.join(P, "stage_constants.json"))
    txt = open(os.path.join(G, "stage_profile_k_step_x.txt")).read()
    head = ("# Round 6, shipped sources + -DTB_PROFILE (tools/gpu_stage_profile.py), headline shape, fp16-pair operands; machine-readable twin: "
            "profiles/stage_constants.json (read by bench.py::structural_floor and tools/gen_warm_schedule.py)\n")
    open(os.path.join(P, "r06_stage_profile_k_step_x.txt"), "w").write(head + "".join(l for l in txt.splitlines(True) if not l.startswith("rc=")))
    # ---- kernel statistics of the end-to-end loops
    parts = ["# rocprofv3 --kernel-trace --stats of tests/probes/gpu_e2e_prefetch_loop.py (tools/gpu_round6_final.sh), end of round 6: WaymoMotion.test_step with a fresh host\n"
             "# batch per call at the headline shape (32 scenes, K = 1, 90 steps), one repetition of the loop after its warm-up.  Before this round's\n"
             "# staging: profiles/r06_rocprof_e2e_before.txt (~90 torch kernels + ~20 pageable copies per step).  Durations in microseconds.\n"]
    for mode, what in (("plain", "plain calls (PREFETCH=0)"), ("lanes", "wm.pipeline(loader, lanes=2) (LANES=2)")):
        f = os.path.join(G, "prof_e2e_" + mode, "e2e_kernel_stats.csv")
        if os.path.exists(f):
            parts.append("## %s\n%s\n" % (what, kernel_table(f)))
    open(os.path.join(P, "r06_rocprof_e2e.txt"), "w").write("\n".join(parts))
    # ---- the end-to-end block as a record of its own (INTEGRATION.md cites it)
    e2e = {"from": "profiles/r06_bench_final.json (`e2e`)", "src_sha256": d.get("src_sha256"), "kernel_only": {"value": d["value"], "ms_per_step": d["ms_per_step"],
           "encode_ms": d.get("encode_ms")}, "e2e": d["e2e"]}
    open(os.path.join(P, "r06_e2e_bench.txt"), "w").write(json.dumps(e2e, indent=1) + "\n")
    # ---- fuzz
    f = os.path.join(ROOT, "gpurun_out", "fuzz_r06.txt")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "r06_fuzz_log.txt"))
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "fp32_exact", d.get("value_fp32_exact"),
          "src", d.get("src_sha256", "")[:16])
    pk = json.load(open(os.path.join(P, "pmc_step_kernel.json")))
    print("pmc src", str(pk.get("src_sha256"))[:16])
    return 0


if __name__ == "__main__":
    sys.exit(main())
