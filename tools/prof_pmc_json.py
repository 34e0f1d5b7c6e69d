"""gpurun_out/pmc_step/<name>/ (tools/gpu_pmc_step.sh) -> the small JSON bench.py reads (profiles/pmc_step_kernel.json): per fused
launch of the step kernel, medians over the launches of the short profiled run, for the headline shape (`fp32`, `bf16`) and the
BASELINE configs[3] / configs[4] shapes (`k6_*`, `stress_*`), stamped with the fingerprint of the sources + flags of the library the passes ran on (and that file's own SHA-256).

Normalisations (checked against each other on this kernel, see `checks`):
  * FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x 2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section);
  * SQ_WAVE_CYCLES counts resident waves in units of FOUR clocks (round 1's launches, 512 waves for the whole launch:
    4 x SQ_WAVE_CYCLES / 512 = kernel-trace duration x ~2.3 GHz); SQ_BUSY_CYCLES is summed over the 32 shader engines
    (/ 32 = the same launch length);
  * SQ_VALU_MFMA_BUSY_CYCLES is a plain clock count summed over SIMDs: / (SIMD clocks that hold a wave) = the fraction of the
    occupied SIMD time during which the matrix pipe executes (matches the instruction count: MFMAs per wave x 16 clocks)."""
import csv
import glob
import hashlib
import json
import os
import statistics
import sys

root = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("TB_HIP_LIB") or os.path.join(ROOT, "trafficbots_amd", "lib", "libtrafficbots_hip.so")
sys.path.insert(0, ROOT)
import __graft_entry__ as _ge  # noqa: E402

# src_sha256 = fingerprint of sources + flags (what bench.py's pmc_matches_build compares: a rebuild of equal sources gives equal device
# code but never an equal .so -- hipcc embeds a fresh __hip_cuid_* per invocation); lib_sha256 is kept as a record of the file profiled
out = {"src_sha256": _ge.build_fingerprint(),
       "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,
       "collected_by": "tools/gpu_pmc_step.sh (rocprofv3 --kernel-trace --stats + separate --pmc passes over short bench.py runs)"}
for name in sorted(os.listdir(root)):
    d = os.path.join(root, name)
    if not os.path.isdir(d) or not os.path.isdir(os.path.join(d, "stats")):
        continue
    # the fused launch is the most frequent variant of k_step_x<PRE = false, ...> in the run (LEAN carve or not, any future tiling)
    dur = {}
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            kn = r.get("Kernel_Name", "")
            if "k_step" in kn and "<true" not in kn.split("k_step", 1)[1][:12]:
                dur.setdefault(kn.split("(")[0], []).append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    if not dur:
        continue
    kern = max(dur, key=lambda k_: sum(dur[k_]))
    v = dur[kern]
    med = statistics.median(v)
    fused = [x for x in v if 0.75 * med < x < 1.5 * med]
    cnt = {}
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r.get("Kernel_Name", "").split("(")[0] != kern:
                continue
            cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    m = {k: statistics.median(x) for k, x in cnt.items()}
    rec = {"kernel": kern.replace("void ", ""), "launches_profiled": len(v),
           "avg_fused_launch_us_kernel_trace": (sum(fused) / len(fused)) if fused else None, "counters_median_per_launch": m}
    if "FETCH_SIZE" in m:
        rec["fetch_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in m:
        rec["write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
    if "SQ_WAVE_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        # launch length in clocks from SQ_BUSY_CYCLES (summed over the 32 shader engines), occupied SIMD time from SQ_WAVE_CYCLES
        # (units of four clocks)
        clocks = m["SQ_BUSY_CYCLES"] / 32.0
        rec["launch_clocks_from_SQ_BUSY_CYCLES"] = clocks
        rec["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"])
        rec["mfma_busy_chip_wide"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * clocks)
        rec["mfma_busy_how"] = ("SQ_VALU_MFMA_BUSY_CYCLES (plain clocks, summed over SIMDs) / (4 x SQ_WAVE_CYCLES) = share of the wave-resident "
                                "SIMD time in which the matrix pipe executes; chip-wide = / (1024 SIMDs x launch "
                                "clocks), launch clocks = SQ_BUSY_CYCLES / 32 shader engines; rocprofv3 --pmc pass of tools/gpu_pmc_step.sh")
        if rec["avg_fused_launch_us_kernel_trace"]:
            rec["checks"] = {"clock_GHz_implied": clocks / (rec["avg_fused_launch_us_kernel_trace"] * 1e3),
                             "simd_occupancy": 4.0 * m["SQ_WAVE_CYCLES"] / (1024 * clocks)}
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        rec["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in m:
        rec["l1_to_l2_read_bytes_per_launch"] = m["TCP_TCC_READ_REQ_sum"] * 128
    # dominant one-time encoder kernel (kernel stats of the same run)
    enc = {}
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            kn = r.get("Name", "")
            if "k_step" in kn or "rollout" in kn or "k_fuse_hoist" in kn or "k_kv_hoist_x(" in kn or "k_pre_replicate" in kn:
                continue
            enc[kn.split("(")[0]] = float(r.get("TotalDurationNs", 0))
    if enc:
        top = max(enc, key=enc.get)
        rec["encode_dominant_kernel"] = f"{top} ({enc[top] / sum(enc.values()):.0%} of the encoder kernels' time)"
    out[name] = rec
print(json.dumps(out, indent=1))
