import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracles work on matrices of a few dozen rows: with torch's default of one thread per core (128 on the GPU box's
    # 256-core host) a 90-step oracle replay takes 7.5 s, with 8 threads 0.6 s (tests/probes/oracle_thread_timing.py).  Only the
    # oracles run on the CPU here (bench.py's cpu_baseline sets its own thread count in its own process).
    import torch

    torch.set_num_threads(min(8, torch.get_num_threads()))


def require_h5():
    """Build the packed-h5 reader if this checkout has not yet (g++, seconds); skip the calling test on a host without HDF5."""
    import __graft_entry__ as ge

    ge.build_h5()
    if not os.path.exists(ge.H5_LIB):
        pytest.skip("no HDF5 headers / runtime on this host: packed-h5 reader not built")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(g["meta_json"]).decode())
    return g, meta


def golden_inputs(meta):
    """Regenerate the exact inputs / weights / latent noise of a golden case from its seeds."""
    from trafficbots_amd import synth
    from trafficbots_amd.config import load_model_config

    cfg = load_model_config(overrides={"time_step_end": meta["time_step_end"], "n_joint_future": meta["k"], **meta.get("overrides", {})})
    sd = synth.case_state_dict(meta)
    batch = synth.make_batch(meta["base_seed"], meta["n_scene"], **meta["scene"])
    n = meta["n_scene"] * meta["k"]
    eps = synth.make_latent_noise(meta["base_seed"] + 99, n, meta["scene"]["n_agent"])
    return cfg, sd, batch, eps


@pytest.fixture(scope="session")
def golden_names():
    return ["c1_plumbing", "small_k1", "masks_k3", "degenerate", "headline_2", "headline_k6"]
