"""GPU probe: the oracle-checked shape fuzzer (tests/probes/gpu_fuzz_shapes.py: random A / P / T / B / K incl. 1, 17, 33, 65, 130 ...)
with every tensor handed to the C ABI between unmapped pages (tests/guard/guard_pool.py) and -- when TB_WS_GUARD is set -- every carve
of the library's workspace as well.  usage: GUARD_AT_END=0|1 TB_WS_GUARD=2|1 FUZZ_SEED=.. [GUARD_FUZZ_SCRIPT=gpu_fuzz_validation.py | gpu_fuzz_rules_post_metrics.py | gpu_fuzz_warm_start.py |
gpu_fuzz_bf16.py] python tests/probes/gpu_guard_fuzz.py <cases>"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "guard"))
import guard_pool  # noqa: E402

pool = guard_pool.install(os.environ.get("GUARD_AT_END", "0") == "1")
try:
    runpy.run_path(os.path.join(ROOT, "tests", "probes", os.environ.get("GUARD_FUZZ_SCRIPT", "gpu_fuzz_shapes.py")), run_name="__main__")
finally:
    n = pool.n_shadow
    guard_pool.uninstall(pool)
print(f"GUARD-FUZZ-OK ({n} guarded buffers, at_end={pool.at_end})")
