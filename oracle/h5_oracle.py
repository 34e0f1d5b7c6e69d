"""TEST INFRASTRUCTURE, not product code: numpy restatement of the reference's packed-h5 datasets for SURVEY 8(f)-4.

Follows `src/data_modules/data_h5_womd.py`: `DatasetBase.__init__` (:10-18), `DatasetTrain.__getitem__` (:27-35, with the episode
index passed in instead of drawn), `DatasetVal.__getitem__` (:38-55) and the default collate of its DataLoader (:229-241).  The
reference reads through h5py, which this image does not have; the restatement reads the same HDF5 objects through the HDF5 C
library directly from Python (ctypes), one call chain per dataset, independent of `trafficbots_amd/csrc/tb_h5_loader.cpp`.

PINNED by the reference itself: tests/golden/gen_h5_reference.py imports the reference's `data_modules.data_h5_womd` in the build
container with an h5py stand-in (`tools/ref_shim.install_h5py`: h5py's File / group / dataset / attrs object model over the `H5Reader`
below, i.e. over the HDF5 C library) and runs ITS `DatasetVal` / `DatasetTrain.__getitem__` + torch's default collate on packed files
written from seeds; dtype / shape / sha256 of every key of those batches are committed as tests/golden/h5_reference.json, and
tests/test_h5_loader.py checks `getitem_*` + `collate` below (and the product reader) against them bit for bit.  What the stand-in
cannot pin is h5py's own dtype mapping (8-bit FALSE/TRUE enum <-> numpy bool, variable-length UTF-8 string attribute <-> str), which
`_numpy_dtype` / `attr` restate from h5py's documented behaviour; the on-disk conventions were checked with `h5dump`.

Only tests/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_ENUM = 0, 1, 3, 8
_lib = None


def _hdf5() -> C.CDLL:
    global _lib
    if _lib is None:
        lib = C.CDLL(os.environ.get("TB_HDF5_LIB", "/opt/conda/lib/libhdf5.so.103"), mode=C.RTLD_GLOBAL)
        hid = C.c_int64
        lib.H5open()
        lib.H5Eset_auto2.argtypes = [hid, C.c_void_p, C.c_void_p]
        lib.H5Eset_auto2(0, None, None)
        for name, args, res in (
            ("H5Fopen", [C.c_char_p, C.c_uint, hid], hid), ("H5Fclose", [hid], C.c_int),
            ("H5Dopen2", [hid, C.c_char_p, hid], hid), ("H5Dclose", [hid], C.c_int), ("H5Dget_space", [hid], hid), ("H5Dget_type", [hid], hid),
            ("H5Dread", [hid, hid, hid, hid, hid, C.c_void_p], C.c_int),
            ("H5Aopen_by_name", [hid, C.c_char_p, C.c_char_p, hid, hid], hid), ("H5Aclose", [hid], C.c_int), ("H5Aget_space", [hid], hid),
            ("H5Aget_type", [hid], hid), ("H5Aread", [hid, hid, C.c_void_p], C.c_int),
            ("H5Sget_simple_extent_ndims", [hid], C.c_int), ("H5Sget_simple_extent_dims", [hid, C.POINTER(C.c_uint64), C.c_void_p], C.c_int),
            ("H5Sclose", [hid], C.c_int), ("H5Tget_class", [hid], C.c_int), ("H5Tget_size", [hid], C.c_size_t), ("H5Tclose", [hid], C.c_int),
            ("H5Tis_variable_str", [hid], C.c_int), ("H5Tget_sign", [hid], C.c_int),
        ):
            getattr(lib, name).argtypes = args
            getattr(lib, name).restype = res
        _lib = lib
    return _lib


def _shape(lib, space) -> Tuple[int, ...]:
    rank = lib.H5Sget_simple_extent_ndims(space)
    dims = (C.c_uint64 * max(rank, 1))()
    lib.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims[:rank])


def _numpy_dtype(lib, ty) -> np.dtype:
    cls, size = lib.H5Tget_class(ty), lib.H5Tget_size(ty)
    if cls == _H5T_ENUM and size == 1:
        return np.dtype(np.bool_)  # h5py: enum {FALSE=0, TRUE=1} over int8 <-> numpy bool
    if cls == _H5T_FLOAT:
        return np.dtype({4: np.float32, 8: np.float64}[size])
    if cls == _H5T_INTEGER:
        signed = lib.H5Tget_sign(ty) != 0
        return np.dtype(f"{'i' if signed else 'u'}{size}")
    raise TypeError(f"unsupported HDF5 class {cls} / size {size}")


class H5Reader:
    def __init__(self, filepath: str) -> None:
        self.lib = _hdf5()
        self.file = self.lib.H5Fopen(filepath.encode(), 0, 0)  # H5F_ACC_RDONLY, H5P_DEFAULT
        if self.file < 0:
            raise OSError(f"cannot open {filepath}")

    def close(self) -> None:
        self.lib.H5Fclose(self.file)

    def dataset(self, path: str) -> np.ndarray:
        """np.ascontiguousarray(hf[path]): the file's own dtype, no conversion (the memory type handed to H5Dread is the stored one)."""
        lib = self.lib
        d = lib.H5Dopen2(self.file, path.encode(), 0)
        if d < 0:
            raise KeyError(path)
        sp, ty = lib.H5Dget_space(d), lib.H5Dget_type(d)
        out = np.empty(_shape(lib, sp), dtype=_numpy_dtype(lib, ty))
        if out.size and lib.H5Dread(d, ty, 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError(f"read of {path} failed")
        lib.H5Tclose(ty), lib.H5Sclose(sp), lib.H5Dclose(d)
        return out

    def attr(self, obj: str, name: str):
        lib = self.lib
        a = lib.H5Aopen_by_name(self.file, obj.encode(), name.encode(), 0, 0)
        if a < 0:
            raise KeyError(f"{obj}@{name}")
        sp, ty = lib.H5Aget_space(a), lib.H5Aget_type(a)
        if lib.H5Tget_class(ty) == _H5T_STRING:
            if lib.H5Tis_variable_str(ty) > 0:
                p = C.c_char_p()
                lib.H5Aread(a, ty, C.byref(p))
                val = p.value.decode()
            else:
                buf = C.create_string_buffer(lib.H5Tget_size(ty) + 1)
                lib.H5Aread(a, ty, buf)
                val = buf.value.decode()
        else:
            arr = np.empty(_shape(lib, sp), dtype=_numpy_dtype(lib, ty))
            lib.H5Aread(a, ty, arr.ctypes.data_as(C.c_void_p))
            val = arr if arr.ndim else arr[()]
        lib.H5Tclose(ty), lib.H5Sclose(sp), lib.H5Aclose(a)
        return val


def dataset_len(filepath: str) -> int:
    """`DatasetBase.__init__`, :14-15."""
    r = H5Reader(filepath)
    n = int(r.attr("/", "data_len"))
    r.close()
    return n


def getitem_train(filepath: str, tensor_size: Dict[str, Tuple[int, ...]], idx: int) -> Dict:
    """`DatasetTrain.__getitem__`, :27-35 (idx given instead of np.random.randint)."""
    r = H5Reader(filepath)
    out = {"episode_idx": idx}
    for k in tensor_size:
        out[k] = np.ascontiguousarray(r.dataset(f"{idx}/{k}"))
    r.close()
    return out


def getitem_val(filepath: str, tensor_size: Dict[str, Tuple[int, ...]], idx: int) -> Dict:
    """`DatasetVal.__getitem__`, :38-55."""
    r = H5Reader(filepath)
    g = str(idx)
    out = {"episode_idx": idx, "scenario_id": r.attr(g, "scenario_id"), "scenario_center": r.attr(g, "scenario_center"),
           "scenario_yaw": r.attr(g, "scenario_yaw"), "with_map": r.attr(g, "with_map")}
    for k, size in tensor_size.items():
        out[k] = np.ascontiguousarray(r.dataset(f"{g}/{k}"))
        if out[k].shape != tuple(size):
            assert "agent" in k
            out[k] = np.ones(size, dtype=out[k].dtype)
    r.close()
    return out


def collate(samples: Sequence[Dict]) -> Dict:
    """torch's default_collate on numpy samples: arrays and numbers are stacked, strings stay a list."""
    out: Dict = {}
    for k in samples[0]:
        vals: List = [s[k] for s in samples]
        out[k] = list(vals) if isinstance(vals[0], str) else np.stack([np.asarray(v) for v in vals])
    return out
