"""Host-side rate of the packed-h5 reader at Waymo sizes (64 agents, 1024 polylines, 40 stop points), beside the per-sample numpy
restatement of the reference's DatasetVal (oracle/h5_oracle.py; the reference itself needs h5py and reads every key of
tensor_size_val, hot path or not).  With a GPU: the same loader feeding validation_step, to see whether reading hides behind it.
    python tests/probes/h5_loader_timing.py [n_episode] [batch]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch

from oracle import h5_oracle
from trafficbots_amd import data_h5, synth

n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 32
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = dict(n_agent=64, n_pl=1024, n_tl=40, p_invalid_agent=0.2, p_late_spawn=0.2, pos_range=140.0)
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    episodes, attrs = synth.make_h5_episodes(500, n_ep, **scene)
    t1 = time.perf_counter()
    data_h5.write_packed_h5(d + "/validation.h5", episodes, attrs)
    t2 = time.perf_counter()
    size = os.path.getsize(d + "/validation.h5")
    print(f"{n_ep} episodes: synth {t1 - t0:.1f} s, write {t2 - t1:.1f} s, file {size / 2**20:.1f} MiB ({size / n_ep / 2**10:.0f} KiB / episode)")
    dm = data_h5.DataH5womd(d, batch_size=bs)
    for k in list(dm.tensor_size_val):
        dm.tensor_size_val[k] = episodes[0][k].shape
    dm.setup("validate")
    loader = dm.val_dataloader()
    for rep in range(2):
        t0 = time.perf_counter()
        nbytes = 0
        for batch in loader:
            nbytes += sum(v.numel() * v.element_size() for k, v in batch.items() if k.startswith("packed/"))
        dt = time.perf_counter() - t0
        print(f"packed loader pass {rep}: {n_ep / dt:.0f} episodes/s ({dt / n_ep * 1e3:.2f} ms / episode, {nbytes / n_ep / 2**10:.0f} KiB decoded / episode)")
    t0 = time.perf_counter()
    n_o = min(n_ep, 8)
    for i in range(n_o):
        h5_oracle.getitem_val(d + "/validation.h5", dm.tensor_size_val, i)
    dt = time.perf_counter() - t0
    print(f"DatasetVal restatement (all {len(dm.tensor_size_val)} keys, one process): {n_o / dt:.0f} episodes/s ({dt / n_o * 1e3:.2f} ms / episode)")
    if torch.cuda.is_available():
        from trafficbots_amd.waymo_motion import WaymoMotion

        wm = WaymoMotion(time_step_end=90, n_joint_future=1)
        wm.load_state_dict(synth.make_state_dict(501))
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for batch in loader:
                out = wm.validation_step(batch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"loader -> validation_step pass {rep}: {dt / n_ep * 1e3:.2f} ms / episode at batch {bs}")
        pre = list(loader)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for batch in pre:
                out = wm.validation_step(batch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"already-read packed batches -> validation_step pass {rep}: {dt / n_ep * 1e3:.2f} ms / episode")
        it, waited = iter(loader), 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while True:
            w0 = time.perf_counter()
            batch = next(it, None)
            waited += time.perf_counter() - w0
            if batch is None:
                break
            out = wm.validation_step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"loader -> validation_step: {dt / n_ep * 1e3:.2f} ms / episode, of which {waited / n_ep * 1e3:.2f} ms waiting for the reader")
        dm.setup("validate")  # a fresh loader: cold chunk index, warm GPU
        it, waited = iter(dm.val_dataloader()), 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while True:
            w0 = time.perf_counter()
            batch = next(it, None)
            waited += time.perf_counter() - w0
            if batch is None:
                break
            out = wm.validation_step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"cold-index loader -> validation_step: {dt / n_ep * 1e3:.2f} ms / episode, of which {waited / n_ep * 1e3:.2f} ms waiting for the reader")
        for label in ("writes", "starts from"):  # a fresh data module per pass, as a new run would have
            dmi = data_h5.DataH5womd(d, batch_size=bs, index_dir=d)
            dmi.tensor_size_val.update(dm.tensor_size_val)
            dmi.setup("validate")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for batch in dmi.val_dataloader():
                out = wm.validation_step(batch)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"fresh loader that {label} the on-disk chunk index -> validation_step: {dt / n_ep * 1e3:.2f} ms / episode")
        print(f"index file: {os.path.getsize(d + '/validation.h5.r0of1.tbidx') / n_ep:.0f} B / episode")
        mem = [{k: np.stack([episodes[i][k] for i in range(j, j + bs)]) for k in episodes[0]} for j in range(0, n_ep, bs)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in mem:
            out = wm.validation_step(b)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"in-memory reference-layout batches -> validation_step: {dt / n_ep * 1e3:.2f} ms / episode")
