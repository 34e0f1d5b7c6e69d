"""SURVEY 8(f)-4, host side: the packed-h5 reader against the numpy restatement of the reference's datasets
(`oracle/h5_oracle.py`), against the content the files were written from, and -- the pin -- both of them against what the REFERENCE's
own dataset classes returned for the same files (tests/golden/h5_reference.json, made by tests/golden/gen_h5_reference.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import h5_oracle
from trafficbots_amd import data_h5, synth
from trafficbots_amd.runtime import gt_from_batch, scene_from_batch

@pytest.fixture(scope="module", autouse=True)
def _reader_built():
    from conftest import require_h5

    require_h5()


SCENE = dict(n_agent=7, n_pl=12, n_tl=5, p_invalid_agent=0.3, p_late_spawn=0.3, p_early_exit=0.2, p_invalid_pl=0.2)
N_NO_SIM = 4


@pytest.fixture(scope="module")
def packed(tmp_path_factory):
    d = tmp_path_factory.mktemp("h5")
    episodes, attrs = synth.make_h5_episodes(4100, 5, **SCENE)
    data_h5.write_packed_h5(str(d / "validation.h5"), episodes, attrs)
    test_keys = [k for k in episodes[0] if k.startswith("history/") or k.startswith("map/")]
    data_h5.write_packed_h5(str(d / "testing.h5"), [{k: e[k] for k in test_keys} for e in episodes], attrs)
    train_keys = None
    dm = data_module(str(d))
    train_keys = list(dm.tensor_size_train)
    data_h5.write_packed_h5(str(d / "training.h5"), [{k: e[k] for k in train_keys} for e in episodes])
    return str(d), episodes, attrs


def data_module(data_dir, **kw):
    dm = data_h5.DataH5womd(data_dir, batch_size=2, n_agent=kw.pop("n_agent", SCENE["n_agent"]), n_pl=SCENE["n_pl"], n_tl_stop=SCENE["n_tl"], **kw)
    # the synthetic files are smaller than Waymo's in the two sizes the reference hard-codes
    for table in (dm.tensor_size_train, dm.tensor_size_test, dm.tensor_size_val):
        for k, size in list(table.items()):
            if "agent_no_sim" in k:
                table[k] = tuple(N_NO_SIM if s == 256 else s for s in size)
            if "tl_lane" in k:
                table[k] = tuple(6 if s == 100 else s for s in size)
    return dm


def test_library_exports_every_declared_symbol():
    lib = data_h5.load()
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "trafficbots_h5.h")).read()
    for name in data_h5.EXPORTS:
        assert name + "(" in header
        getattr(lib, name)
    import re
    declared = set(re.findall(r"\b(tb_h5_\w+)\(", header))
    assert declared == set(data_h5.EXPORTS)


def test_file_round_trip_is_exact(packed):
    d, episodes, attrs = packed
    assert h5_oracle.dataset_len(d + "/validation.h5") == 5
    f = data_h5.PackedH5File(d + "/validation.h5")
    assert len(f) == 5
    for i in (0, 4):
        a = f.episode_attrs(i)
        assert a["scenario_id"] == attrs[i]["scenario_id"] and a["with_map"] == attrs[i]["with_map"]
        assert np.array_equal(a["scenario_center"], attrs[i]["scenario_center"]) and a["scenario_yaw"] == attrs[i]["scenario_yaw"]
        sample = h5_oracle.getitem_val(d + "/validation.h5", {k: v.shape for k, v in episodes[i].items()}, i)
        for k, v in episodes[i].items():
            assert sample[k].dtype == v.dtype and np.array_equal(sample[k], v), k
            assert f.dataset_shape(i, k) == v.shape


@pytest.mark.parametrize("split", ["val", "test", "train"])
def test_reference_batch_matches_dataset_restatement(packed, split):
    d, _, _ = packed
    dm = data_module(d)
    path, table = {"val": (dm.path_val_h5, dm.tensor_size_val), "test": (dm.path_test_h5, dm.tensor_size_test),
                   "train": (dm.path_train_h5, dm.tensor_size_train)}[split]
    idx = [3, 0, 4]
    get = h5_oracle.getitem_train if split == "train" else h5_oracle.getitem_val
    want = h5_oracle.collate([get(path, table, i) for i in idx])
    got = data_h5.PackedH5File(path).read_reference_batch(idx, table, with_attrs=split != "train")
    assert set(got) == set(want)
    for k, w in want.items():
        if isinstance(w, list):
            assert got[k] == w
            continue
        g = got[k].numpy()
        assert g.dtype == w.dtype and g.shape == w.shape and np.array_equal(g, w), k


@pytest.mark.parametrize("split", ["val", "test", "train"])
def test_packed_batch_equals_preprocessing_of_the_reference_batch(packed, split):
    """decode-in-the-reader == `scene_from_batch` / `gt_from_batch` applied to what the reference's DataLoader would yield"""
    d, _, _ = packed
    dm = data_module(d)
    dm.setup({"val": "validate", "test": "test", "train": "fit"}[split])
    loader = {"val": dm.val_dataloader, "test": dm.test_dataloader, "train": dm.train_dataloader}[split]()
    assert len(loader) == 3
    path, table = (loader.filepath, loader.tensor_size)
    get = h5_oracle.getitem_train if split == "train" else h5_oracle.getitem_val
    n_seen = 0
    for batch in loader:
        idx = batch["episode_idx"].tolist()
        n_seen += len(idx)
        ref = h5_oracle.collate([get(path, table, i) for i in idx])
        if split == "train":  # scene_centric.py:103-133: the history is the first 11 steps of the episode
            for k in list(ref):
                if k.startswith("agent/") or k.startswith("tl_stop/"):
                    ref["history/" + k] = ref[k][:, :11] if k.rsplit("/", 1)[-1] in data_h5._STEP_KEYS else ref[k]
        else:
            assert batch["scenario_id"] == ref["scenario_id"]
            assert np.array_equal(batch["scenario_center"].numpy(), ref["scenario_center"])
            assert np.array_equal(batch["with_map"].numpy(), ref["with_map"])
        want = scene_from_batch(ref, "cpu", with_gt=split != "test")  # (the staged path: host-made extras included)
        if split != "test":  # ... whose ground-truth part equals the torch-op conversion
            for k, v in gt_from_batch(ref, "cpu").items():
                assert (want["gt"][k] == v) if isinstance(v, bool) else torch.equal(want["gt"][k], v), k
        got = data_h5.scene_from_packed(batch, "cpu")

        def same(a, b, where):
            assert set(a) == set(b), (where, set(a) ^ set(b))
            for k in a:
                if isinstance(a[k], dict):
                    same(a[k], b[k], where + k + "/")
                elif isinstance(a[k], (bool, tuple)):
                    assert a[k] == b[k], where + k
                else:
                    assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k], b[k]), where + k

        same(want, got, "")
    assert n_seen == 5
    if split != "train":
        assert sorted(sum((b["episode_idx"].tolist() for b in loader), [])) == list(range(5))  # re-iterable


@pytest.mark.parametrize("storage", [dict(chunk_div=2), dict(chunk_div=3, shuffle=False), dict(chunk_div=2, deflate=0), dict(chunk_div=0, deflate=0),
                                     dict(chunk_div=1, deflate=9)])
def test_storage_forms_and_reader_paths_agree(packed, tmp_path, storage):
    """multi-chunk tensors with partial edge chunks (what h5py's automatic chunking produces), unshuffled / uncompressed / contiguous
    storage: the threaded chunk decoder, the single-threaded one and the H5Dread path (n_threads = 0) give the stored content"""
    _, episodes, attrs = packed
    path = str(tmp_path / "v.h5")
    data_h5.write_packed_h5(path, episodes, attrs, **storage)
    dm = data_module(str(tmp_path))
    f = data_h5.PackedH5File(path)
    idx = [4, 1, 2]
    specs = [(k, size, data_h5.reference_kind(k), 0, False) for k, size in dm.tensor_size_val.items()]
    specs += [("agent/pos", dm.tensor_size_val["agent/pos"], data_h5.F32, 11, False), ("tl_stop/state", dm.tensor_size_val["tl_stop/state"], data_h5.ONEHOT_I32, 47, False),
              ("agent/valid", dm.tensor_size_val["agent/valid"], data_h5.MASK_U8, 46, False), ("map/type", dm.tensor_size_val["map/type"], data_h5.ONEHOT_I32, 0, False)]
    a, b, c = (f.read_keys(idx, specs, n_threads=n) for n in (8, 1, 0))
    for (key, size, kind, lead, _), ta, tb, tc in zip(specs, a, b, c):
        assert torch.equal(ta, tb) and torch.equal(ta, tc), key
        want = np.stack([episodes[i][key] for i in idx])
        if lead:
            want = want[:, :lead]
        if kind == data_h5.ONEHOT_I32:
            want = np.where(want.any(-1), want.argmax(-1), -1).astype(np.int32)
        assert np.array_equal(ta.numpy(), want.astype(ta.numpy().dtype)), key


def test_chunk_index_makes_second_visit_metadata_free(packed, capfd):
    d, episodes, _ = packed
    dm = data_module(d)
    f = data_h5.PackedH5File(dm.path_val_h5)
    specs = [(k, size, data_h5.reference_kind(k), 0, False) for k, size in dm.tensor_size_val.items()]
    os.environ["TB_H5_DEBUG"] = "1"
    try:
        first = f.read_keys([0, 3], specs)
        second = f.read_keys([3, 0], specs)
        f.set_index_cache(0)
        third = f.read_keys([0, 3], specs)
    finally:
        del os.environ["TB_H5_DEBUG"]
    log = [l for l in capfd.readouterr().err.splitlines() if "tb_h5_read_batch" in l]
    n = 2 * len(specs)
    assert f"({n} probed" in log[0] and "(0 probed" in log[1] and f"({n} probed" in log[2], log
    for a, b, c in zip(first, second, third):
        assert torch.equal(a, b.flip(0)) and torch.equal(a, c)


def test_random_shapes_chunkings_and_slices(tmp_path):
    """seeded sweep: ranks 1-4, dims 1-9 (class dims 1-6), every kind, chunk splits 1-4 per dim, leading-row slices"""
    rng = np.random.default_rng(77)
    for trial in range(12):
        episodes, specs, want = [dict() for _ in range(3)], [], []
        for j in range(10):
            rank = int(rng.integers(1, 5))
            dims = tuple(int(x) for x in rng.integers(1, 10, rank))
            kind = int(rng.integers(0, 4))
            key = f"g{j % 3}/agent_t{j}"
            for ep in episodes:
                if kind == data_h5.F32:
                    ep[key] = rng.normal(size=dims).astype(np.float32)
                elif kind == data_h5.I64:
                    ep[key] = rng.integers(-2**40, 2**40, dims).astype(np.int64)
                else:
                    ep[key] = rng.random(dims) < (0.15 if kind == data_h5.ONEHOT_I32 else 0.5)
            lead = int(rng.integers(0, dims[0] + 1)) if (rank > 1 or kind != data_h5.ONEHOT_I32) and rng.random() < 0.5 else 0
            specs.append((key, dims, kind, lead, False))
        path = str(tmp_path / f"r{trial}.h5")
        data_h5.write_packed_h5(path, episodes, chunk_div=int(rng.integers(0, 5)), deflate=int(rng.choice([0, 1, 4])), shuffle=bool(rng.integers(0, 2)))
        f = data_h5.PackedH5File(path)
        idx = [2, 0, 1, 2]
        for n_threads in (4, 0):
            got = f.read_keys(idx, specs, n_threads=n_threads)
            for (key, dims, kind, lead, _), t in zip(specs, got):
                w = np.stack([episodes[i][key] for i in idx])
                if lead:
                    w = w[:, :lead]
                if kind == data_h5.ONEHOT_I32:
                    w = np.where(w.any(-1), w.argmax(-1), -1).astype(np.int32)
                assert t.shape == w.shape and np.array_equal(t.numpy(), w.astype(t.numpy().dtype)), (trial, key, dims, kind, lead, n_threads)


def test_chunk_index_on_disk(packed, tmp_path, capfd):
    d, _, _ = packed
    idx_dir = str(tmp_path)

    def run():
        dm = data_module(d, index_dir=idx_dir)
        dm.setup("validate")
        os.environ["TB_H5_DEBUG"] = "1"
        try:
            out = list(dm.val_dataloader())
        finally:
            del os.environ["TB_H5_DEBUG"]
        return out, [l for l in capfd.readouterr().err.splitlines() if "tb_h5_read_batch" in l]

    first, log1 = run()
    assert os.path.exists(idx_dir + "/validation.h5.r0of1.tbidx") and all("(0 probed" not in l for l in log1)
    second, log2 = run()  # a new process would do the same: every handle starts from the file's index
    assert len(log2) == 3 and all("(0 probed" in l for l in log2), log2
    for a, b in zip(first, second):
        assert all(torch.equal(a[k], b[k]) for k in a if torch.is_tensor(a[k]) and k != "packed/_slab")  # (its alignment padding is not written)
    # an index made for another version of the data file is refused, loudly at the C level and quietly by the loader
    f = data_h5.PackedH5File(d + "/testing.h5")
    assert not f.load_index(idx_dir + "/validation.h5.r0of1.tbidx") and "another version" in data_h5.load().tb_h5_last_error().decode()
    with open(idx_dir + "/bad.tbidx", "wb") as fh:
        fh.write(b"nonsense")
    assert not f.load_index(idx_dir + "/bad.tbidx") and "not an index file" in data_h5.load().tb_h5_last_error().decode()


def test_agent_count_override_yields_dummy_agents(packed):
    """`data_h5_womd.py:50-52`: an agent tensor whose stored shape differs from the configured one is replaced by ones"""
    d, _, _ = packed
    dm = data_module(d, n_agent=9)
    ref = h5_oracle.collate([h5_oracle.getitem_val(dm.path_test_h5, dm.tensor_size_test, i) for i in (1, 2)])
    got = data_h5.PackedH5File(dm.path_test_h5).read_reference_batch([1, 2], dm.tensor_size_test, with_attrs=True)
    for k, w in ref.items():
        if not isinstance(w, list):
            assert np.array_equal(got[k].numpy(), w), k
    assert got["history/agent/pos"].shape == (2, 11, 9, 2) and bool((got["history/agent/pos"] == 1).all())
    packed_batch = data_h5.PackedH5File(dm.path_test_h5).read_packed_batch([1, 2], dm.tensor_size_test, "test")
    assert bool((packed_batch["packed/agent_type"] == 0).all()) and bool((packed_batch["packed/agent_valid"] == 1).all())
    assert packed_batch["packed/map_pos"].shape == (2, 12, 20, 2)


def test_sharded_loaders_partition_the_file(packed):
    d, _, _ = packed
    seen = []
    for r in range(2):
        dm = data_module(d, rank=r, world_size=2)
        dm.setup("validate")
        seen.append(sum((b["episode_idx"].tolist() for b in dm.val_dataloader()), []))
    assert seen == [[0, 2, 4], [1, 3]]


def test_errors_are_loud(packed, tmp_path):
    d, _, _ = packed
    with pytest.raises(RuntimeError, match="cannot open"):
        data_h5.PackedH5File(str(tmp_path / "missing.h5"))
    f = data_h5.PackedH5File(d + "/testing.h5")
    with pytest.raises(RuntimeError, match="no dataset"):
        f.read_key([0], "agent/goal", (7, 4), data_h5.F32)  # a testing file holds no ground truth
    with pytest.raises(RuntimeError, match="differs"):
        f.read_key([0], "map/pos", (13, 20, 2), data_h5.F32)  # only agent tensors may be replaced by dummies
    with pytest.raises(RuntimeError, match="no dataset"):
        f.read_key([7], "map/pos", (12, 20, 2), data_h5.F32)


def _digest(v):
    import hashlib

    if isinstance(v, (list, tuple)) and v and isinstance(v[0], str):
        return {"kind": "str", "values": list(v)}
    a = np.ascontiguousarray(v.numpy() if torch.is_tensor(v) else np.asarray(v))
    return {"kind": "array", "dtype": str(a.dtype), "shape": list(a.shape), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}


def test_reader_and_restatement_match_the_references_own_datasets(tmp_path):
    """The pin of this row.  tests/golden/h5_reference.json holds, per split and key, dtype / shape / sha256 of the batches the
    REFERENCE's `DatasetVal` / `DatasetTrain.__getitem__` + default collate produced (`data_h5_womd.py:9-55`, run in the build
    container over an h5py stand-in on the HDF5 C library) for files written from seeds.  The same files are rebuilt here; the numpy
    restatement (`oracle/h5_oracle.py`) and the product reader (`tb_h5_read_batch` via `read_reference_batch`) must reproduce every
    entry bit for bit -- including the all-ones dummy tensors when n_agent differs from the file's (:50-53), the episode attributes
    and their dtypes (variable-length string, float64 centre / yaw, bool with_map)."""
    import importlib.util
    import json

    from conftest import GOLDEN_DIR

    spec = importlib.util.spec_from_file_location("gen_h5_reference", os.path.join(GOLDEN_DIR, "gen_h5_reference.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    golden = json.load(open(os.path.join(GOLDEN_DIR, "h5_reference.json")))
    assert golden["case"] == json.loads(json.dumps(gen.CASE)), "the generator's case changed: regenerate the golden"
    case = golden["case"]
    d = str(tmp_path)
    gen.write_files(d, case)
    idx = case["batch"]
    n_checked = 0
    for n_agent, tag in ((case["scene"]["n_agent"], "file_agents"), (case["n_agent_dummy"], "dummy_agents")):
        dm = data_h5.DataH5womd(d, n_agent=n_agent, n_pl=case["scene"]["n_pl"], n_tl_stop=case["scene"]["n_tl"])
        gen.patch_sizes(dm, case)
        for split, path, table in (("val", dm.path_val_h5, dm.tensor_size_val), ("test", dm.path_test_h5, dm.tensor_size_test),
                                   ("train", dm.path_train_h5, dm.tensor_size_train)):
            want = golden["splits"].get(f"{split}/{tag}")
            if want is None:
                continue
            get = h5_oracle.getitem_train if split == "train" else h5_oracle.getitem_val
            restated = h5_oracle.collate([get(path, table, i) for i in idx])
            product = data_h5.PackedH5File(path).read_reference_batch(idx, table, with_attrs=split != "train")
            assert set(restated) == set(want) == set(product), (split, tag, set(want) ^ set(product))
            for k, w in want.items():
                assert _digest(restated[k]) == w, ("restatement", split, tag, k)
                assert _digest(product[k]) == w, ("product reader", split, tag, k)
                n_checked += 1
    assert n_checked == 69 + 38 + 27 + 69 + 38


def _run(cmd):
    import subprocess

    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout}"


def test_reader_on_files_made_by_the_hdf5_command_line_tools(tmp_path):
    """An INDEPENDENT writer: until here every file the reader saw came out of `tb_h5_writer_*`, so reader and
    writer were each other's only witness.  This file is built with the HDF5 project's own tools -- `h5import` turns raw binary
    arrays into datasets of `<episode>/<key>`, `h5repack` re-stores them chunked with the shuffle + gzip filters the packer uses
    (`pack_h5_womd.py:378-392` -> h5py `compression="gzip", shuffle=True`) -- and must decode to the bytes it was made from, through
    the reference-layout path and through the decoded (`tb_scene` layout) path; the repacked chunked file and the contiguous one agree.
    Booleans are 8-bit integers here (h5import cannot make h5py's enum type; that convention is covered by the h5dump checks)."""
    import shutil

    tools = {t: shutil.which(t) or os.path.join(os.environ.get("TB_HDF5_PREFIX", "/opt/conda"), "bin", t) for t in ("h5import", "h5repack", "h5dump")}
    if not all(os.path.exists(p) for p in tools.values()):
        pytest.skip("HDF5 command-line tools not installed")
    episodes, _ = synth.make_h5_episodes(4300, 3, **SCENE)
    keys = ["history/agent/pos", "history/agent/valid", "history/agent/type", "history/agent/object_id", "map/pos", "map/valid",
            "map/type", "history/tl_stop/state", "history/tl_stop/valid", "history/agent/yaw_bbox"]
    plain = str(tmp_path / "cli_plain.h5")
    for i, ep in enumerate(episodes):
        for k in keys:
            v = np.ascontiguousarray(ep[k])
            raw = tmp_path / "raw.bin"
            (v.view(np.uint8) if v.dtype == np.bool_ else v).tofile(raw)
            cls, bits = {"float32": ("FP", 32), "int64": ("IN", 64), "bool": ("UIN", 8)}[str(v.dtype)]
            cfg = tmp_path / "cfg.txt"
            cfg.write_text(f"PATH {i}/{k}\nINPUT-CLASS {cls}\nINPUT-SIZE {bits}\nOUTPUT-CLASS {cls}\nOUTPUT-SIZE {bits}\n"
                           f"RANK {v.ndim}\nDIMENSION-SIZES {' '.join(str(s) for s in v.shape)}\nOUTPUT-ARCHITECTURE NATIVE\n")
            _run([tools["h5import"], str(raw), "-c", str(cfg), "-o", plain])
    packed_like = str(tmp_path / "cli_gzip.h5")
    _run([tools["h5repack"], "-f", "SHUF", "-f", "GZIP=4", plain, packed_like])
    import subprocess
    dump = subprocess.run([tools["h5dump"], "-pH", "-d", "/1/map/pos", packed_like], stdout=subprocess.PIPE, text=True).stdout
    assert "CHUNKED" in dump and "SHUFFLE" in dump and "DEFLATE" in dump, dump  # (the repacked file really carries the filters)
    for path in (plain, packed_like):
        f = data_h5.PackedH5File(path)
        for k in keys:
            want = np.stack([ep[k] for ep in episodes])
            kind = data_h5.reference_kind(k)
            got = f.read_key([0, 1, 2], k, tuple(want.shape[1:]), kind).numpy()
            got = got.view(np.bool_) if want.dtype == np.bool_ else got
            assert got.dtype == want.dtype and np.array_equal(got, want), (path, k)
        # the decoded forms of the hot path: one-hot -> class index, history slice of a longer tensor, squeezed yaw
        ty = f.read_key([2, 0], "history/agent/type", (SCENE["n_agent"], 3), data_h5.ONEHOT_I32).numpy()
        want_ty = np.stack([np.where(episodes[e]["history/agent/type"].any(-1), episodes[e]["history/agent/type"].argmax(-1), -1) for e in (2, 0)])
        assert np.array_equal(ty, want_ty)
        lead = f.read_key([1], "history/agent/pos", episodes[1]["history/agent/pos"].shape, data_h5.F32, n_lead=4).numpy()
        assert np.array_equal(lead[0], episodes[1]["history/agent/pos"][:4])
        yaw = f.read_key([1], "history/agent/yaw_bbox", episodes[1]["history/agent/yaw_bbox"].shape, data_h5.F32, squeeze=True).numpy()
        assert np.array_equal(yaw[0], episodes[1]["history/agent/yaw_bbox"][..., 0])
