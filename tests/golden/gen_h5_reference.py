"""Generator of tests/golden/h5_reference.json (build container only: `python tests/golden/gen_h5_reference.py`).

Pins SURVEY 8(f)-4 with THE REFERENCE ITSELF: the reference's dataset classes (`src/data_modules/data_h5_womd.py:9-55`:
`DatasetVal.__getitem__`, `DatasetTrain.__getitem__`) and torch's default collate (the collate_fn of its DataLoaders, `:229-241`) run
here on packed-h5 files written from seeds (`synth.make_h5_episodes` + `tb_h5_writer_*`), over an h5py stand-in
(`tools/ref_shim.install_h5py`: h5py's File / group / dataset / attrs objects on top of the ctypes -> libhdf5 reader -- the image has
the HDF5 C library, not h5py).  Stored per split and batch: for every key of the collated batch its dtype, shape and the sha256 of
its bytes (strings: the values).  tests/test_h5_loader.py rebuilds the same files and checks `oracle/h5_oracle.py` AND the product
reader (`tb_h5_read_batch` through `DataH5womd.read_reference_batch`) against these."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CASE = dict(seed=4100, n_episode=5, scene=dict(n_agent=7, n_pl=12, n_tl=5, p_invalid_agent=0.3, p_late_spawn=0.3, p_early_exit=0.2,
                                               p_invalid_pl=0.2), n_no_sim=4, n_tl_lane=6, batch=[0, 3, 4], n_agent_dummy=9)


def digest(v):
    if isinstance(v, (list, tuple)) and v and isinstance(v[0], str):
        return {"kind": "str", "values": list(v)}
    a = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    a = np.ascontiguousarray(a)
    return {"kind": "array", "dtype": str(a.dtype), "shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}


def write_files(d, case=CASE):
    from trafficbots_amd import data_h5, synth

    episodes, attrs = synth.make_h5_episodes(case["seed"], case["n_episode"], **case["scene"])
    dm = data_h5.DataH5womd(d, n_agent=case["scene"]["n_agent"], n_pl=case["scene"]["n_pl"], n_tl_stop=case["scene"]["n_tl"])
    data_h5.write_packed_h5(f"{d}/validation.h5", episodes, attrs)
    data_h5.write_packed_h5(f"{d}/testing.h5", [{k: e[k] for k in dm.tensor_size_test} for e in episodes], attrs)
    data_h5.write_packed_h5(f"{d}/training.h5", [{k: e[k] for k in dm.tensor_size_train} for e in episodes])


def patch_sizes(dm, case=CASE):
    """the reference hard-codes Waymo's sizes (`data_h5_womd.py:78-84`): bring the three it cannot be told to the synthetic files'"""
    sc = case["scene"]
    for table in (dm.tensor_size_train, dm.tensor_size_test, dm.tensor_size_val):
        for k, size in list(table.items()):
            size = tuple(size)
            if "agent_no_sim" in k:
                size = tuple(case["n_no_sim"] if s == 256 else s for s in size)
            if "tl_lane" in k:
                size = tuple(case["n_tl_lane"] if s == 100 else s for s in size)
            if k.startswith("map/") or k.startswith("history/map"):
                size = tuple(sc["n_pl"] if s == 1024 else s for s in size)
            if "tl_stop" in k:
                size = tuple(sc["n_tl"] if s == 40 else s for s in size)
            table[k] = size


def main():
    import ref_shim
    from oracle.h5_oracle import H5Reader

    ref = ref_shim.import_reference_datasets(H5Reader)
    from torch.utils.data import default_collate

    out = {"case": CASE, "splits": {}}
    with tempfile.TemporaryDirectory() as d:
        write_files(d)
        for n_agent, tag in ((CASE["scene"]["n_agent"], "file_agents"), (CASE["n_agent_dummy"], "dummy_agents")):
            dm = ref.DataH5womd(d, n_agent=n_agent)
            patch_sizes(dm)
            for split, path, table, cls in (("val", dm.path_val_h5, dm.tensor_size_val, ref.DatasetVal),
                                            ("test", dm.path_test_h5, dm.tensor_size_test, ref.DatasetVal),
                                            ("train", dm.path_train_h5, dm.tensor_size_train, ref.DatasetTrain)):
                if split == "train" and tag == "dummy_agents":
                    continue  # (DatasetTrain has no dummy rule, :27-35)
                ds = cls(path, table)
                assert len(ds) == CASE["n_episode"]
                samples = []
                for i in CASE["batch"]:
                    if split == "train":  # DatasetTrain draws its own index: pin the draw
                        orig = np.random.randint
                        np.random.randint = lambda *_a, _i=i, **_k: _i
                        try:
                            samples.append(ds[0])
                        finally:
                            np.random.randint = orig
                    else:
                        samples.append(ds[i])
                batch = default_collate(samples)
                out["splits"][f"{split}/{tag}"] = {k: digest(v) for k, v in batch.items()}
    with open(os.path.join(HERE, "h5_reference.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out["splits"].items()})


if __name__ == "__main__":
    main()
