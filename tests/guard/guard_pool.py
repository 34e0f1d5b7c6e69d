"""Test infrastructure: the object behind trafficbots_amd.hip.guard_hook (tests/probes/gpu_guard_pages.py, gpu_guard_fuzz.py) -- every
tensor handed to the C ABI is replaced by a copy in a buffer with unmapped address space on both sides (tests/guard/tb_guard.cpp)."""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _flat_bytes(t):
    """the memory of a contiguous tensor as a flat uint8 view (is_contiguous() allows any stride on a size-1 dimension: as_strided)"""
    return torch.as_strided(t, (t.numel(),), (1,), t.storage_offset()).view(torch.uint8)


class _Raw:
    def __init__(self, p, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (p, False), "version": 2}


class GuardPool:
    """hip.guard_hook: shadow(tensor) -> address of a guarded copy; writeback() copies the shadows of the call back."""

    def __init__(self, at_end: bool):
        self.lib = C.CDLL(os.path.join(ROOT, "tests", "guard", "libtb_guard.so"))
        self.lib.tbg_alloc.restype = C.c_void_p
        self.lib.tbg_alloc.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        self.lib.tbg_free.argtypes = [C.c_void_p, C.c_size_t]
        self.lib.tbg_granularity.restype = C.c_size_t
        self.at_end, self.pending, self.live, self.n_shadow = int(at_end), {}, [], 0
        assert self.lib.tbg_granularity() > 0, "hipMemGetAllocationGranularity failed"

    def shadow(self, t):
        key = (t.data_ptr(), t.numel() * t.element_size())
        if key in self.pending:
            return self.pending[key][0]
        n = key[1]
        base, span = C.c_void_p(), C.c_size_t()
        p = self.lib.tbg_alloc(n, self.at_end, C.byref(base), C.byref(span))
        assert p, "tbg_alloc failed"
        self.live.append((base.value, span.value))
        g = torch.as_tensor(_Raw(p, max(n, 1)), device="cuda")[:n] if n else None
        if n:
            g.copy_(_flat_bytes(t))
        self.pending[key] = (p, g, t)
        self.n_shadow += 1
        return p

    def writeback(self):
        for p, g, t in self.pending.values():
            if g is not None:
                _flat_bytes(t).copy_(g)
        self.pending = {}

    def close(self):
        torch.cuda.synchronize()
        for base, span in self.live:
            self.lib.tbg_free(base, span)
        self.live = []




def install(at_end: bool) -> "GuardPool":
    from trafficbots_amd import hip

    pool = GuardPool(at_end)
    hip.guard_hook = pool
    return pool


def uninstall(pool: "GuardPool") -> None:
    from trafficbots_amd import hip

    hip.guard_hook = None
    pool.close()
