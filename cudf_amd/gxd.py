"""ctypes binding of the sharded operators (include/cudf_amd/gxd.h, implemented in C++ over RCCL in
cudf_amd/cpp/src/distributed.cpp -> cudf_amd/libcudf.so).  Python is a thin caller here: it creates the communicator
(the 128-byte RCCL id travels through torch.distributed, which the launcher has set up anyway), hands over device
pointers of torch tensors and wraps the result buffers the C++ side asked its allocator callback for.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from .column import gx_dtype

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libcudf.so")
if not os.path.exists(_PATH):  # no fallback: the sharded operators ARE this library
    raise ImportError(f"{_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
_lib = ctypes.CDLL(_PATH)

_ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
_p, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
_lib.gxd_last_error.restype = ctypes.c_char_p
_lib.gxd_unique_id.argtypes = [_p]
_lib.gxd_comm_create.argtypes = [_p, _i, _i, ctypes.POINTER(_p)]
_lib.gxd_comm_create_loopback.argtypes = [_i, ctypes.POINTER(_p)]
_lib.gxd_comm_destroy.argtypes = [_p]
_lib.gxd_comm_abort.argtypes = [_p]
_lib.gxd_last_timing.argtypes = [_p, ctypes.POINTER(ctypes.c_double)]
_lib.gxd_sort.argtypes = [_p, _i, _p, _i64, _i, _i, _ALLOC, _p, ctypes.POINTER(_p), ctypes.POINTER(_i64), _p]
_lib.gxd_join_build.argtypes = [_p, _i, _p, _i64, _i, _p, ctypes.POINTER(_p)]
_lib.gxd_join_probe.argtypes = [_p, _p, _i64, _i, _ALLOC, _p, ctypes.POINTER(_p), ctypes.POINTER(_p), ctypes.POINTER(_i64), _p]
_lib.gxd_join_destroy.argtypes = [_p]
_lib.gxd_test_set_slot_scale.argtypes = [ctypes.c_double]
_lib.gxd_test_set_slot_scale.restype = None
_lib.gxd_test_set_row_bits.argtypes = [_i]
_lib.gxd_test_set_row_bits.restype = None
_lib.gxd_test_set_sort_mode.argtypes = [_i]
_lib.gxd_test_set_sort_mode.restype = None
_lib.gxd_groupby_sum_count.argtypes = [_p, _i, _p, _i, _p, _i64, _i64, _i, _ALLOC, _p, ctypes.POINTER(_p), ctypes.POINTER(_p),
                                       ctypes.POINTER(_p), ctypes.POINTER(_i64), _p]


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: {rc}: {_lib.gxd_last_error().decode()}")


class _Results:
    """the allocator the C++ side calls for its RESULT buffers: torch owns them"""

    def __init__(self):
        self.bufs: List[torch.Tensor] = []

        def alloc(nbytes, _ctx):
            t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device="cuda")
            self.bufs.append(t)
            return t.data_ptr()
        self.cb = _ALLOC(alloc)

    def release(self):
        """break the cycle self -> cb -> alloc closure -> self: without it the result buffers of every call stay alive
        until the cyclic garbage collector runs (8 GB per 1e9-row sort: profiles/r3_run14_alloc_probe.txt)"""
        self.bufs = []
        self.cb = None

    def take(self, ptr, count: int, dtype: torch.dtype) -> torch.Tensor:
        if count == 0 or not ptr:
            return torch.empty(0, dtype=dtype, device="cuda")
        for t in self.bufs:
            if t.data_ptr() == ptr:
                return t[: count * torch.empty(0, dtype=dtype).element_size()].view(dtype)
        raise RuntimeError("gxd: result pointer was not produced by the allocator")


def set_slot_scale(scale: float):
    """test hook: see gxd_test_set_slot_scale"""
    _lib.gxd_test_set_slot_scale(float(scale))


def set_row_bits(bits: int):
    """test hook: see gxd_test_set_row_bits"""
    _lib.gxd_test_set_row_bits(int(bits))


def set_sort_mode(mode: int):
    """test hook: see gxd_test_set_sort_mode"""
    _lib.gxd_test_set_sort_mode(int(mode))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _np_dtype(t: torch.Tensor):
    return np.dtype(str(t.dtype).replace("torch.", ""))


class Communicator:
    """gxd_comm: an RCCL communicator of its own (next to torch.distributed's), an exchange stream and the persistent
    partition / receive buffers of the sharded operators."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        idbuf = (ctypes.c_char * 128)()
        if self.rank == 0:
            _check(_lib.gxd_unique_id(idbuf), "gxd_unique_id")
        if self.world > 1:
            t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).cuda()
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            idbuf = (ctypes.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        self._h = ctypes.c_void_p()
        _check(_lib.gxd_comm_create(idbuf, self.world, self.rank, ctypes.byref(self._h)), "gxd_comm_create")

    @classmethod
    def loopback(cls, world: int) -> "List[Communicator]":
        """`world` logical ranks on the current device (gxd_comm_create_loopback): device-to-device copies stand in for RCCL.
        Each communicator must be driven by its own host thread on its own non-blocking stream -- see `run_ranks`."""
        hs = (_p * world)()
        _check(_lib.gxd_comm_create_loopback(world, hs), "gxd_comm_create_loopback")
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c.rank, c.world, c._h = r, world, ctypes.c_void_p(hs[r])
            out.append(c)
        return out

    def close(self):
        if self._h:
            _lib.gxd_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def abort(self):
        """gxd_comm_abort: callable from another thread while an operator of this communicator is blocked in a collective"""
        if self._h:
            _lib.gxd_comm_abort(self._h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_timing(self) -> Tuple[float, float, float]:
        ms = (ctypes.c_double * 3)()
        _lib.gxd_last_timing(self._h, ms)
        return tuple(ms)

    # ---- operators: this rank's shard in, this rank's shard of the result out
    def sort(self, keys: torch.Tensor, chunks: int = 0, force_exchange: bool = False) -> torch.Tensor:
        keys = keys.contiguous()
        res = _Results()
        out, n = ctypes.c_void_p(), ctypes.c_int64()
        try:
            _check(_lib.gxd_sort(self._h, gx_dtype(_np_dtype(keys)), keys.data_ptr(), keys.numel(), chunks, int(force_exchange), res.cb, None,
                                 ctypes.byref(out), ctypes.byref(n), _stream()), "gxd_sort")
            return res.take(out.value, n.value, keys.dtype)
        finally:
            res.release()

    def groupby_sum_count(self, keys: torch.Tensor, vals: torch.Tensor, max_groups: int = 0, force_exchange: bool = False):
        keys, vals = keys.contiguous(), vals.contiguous()
        res = _Results()
        ok, os_, oc, g = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        try:
            _check(_lib.gxd_groupby_sum_count(self._h, gx_dtype(_np_dtype(keys)), keys.data_ptr(), gx_dtype(_np_dtype(vals)), vals.data_ptr(),
                                              keys.numel(), max_groups, int(force_exchange), res.cb, None, ctypes.byref(ok), ctypes.byref(os_),
                                              ctypes.byref(oc), ctypes.byref(g), _stream()), "gxd_groupby_sum_count")
            sdt = torch.float64 if vals.dtype.is_floating_point else torch.int64
            return res.take(ok.value, g.value, keys.dtype), res.take(os_.value, g.value, sdt), res.take(oc.value, g.value, torch.int64)
        finally:
            res.release()


class HashJoin:
    """gxd_join: cudf::hash_join over sharded tables -- build side exchanged and hashed once, probed many times"""

    def __init__(self, comm: Communicator, build_keys: torch.Tensor, force_exchange: bool = False):
        self.comm = comm
        self._keys = build_keys.contiguous()
        self._h = ctypes.c_void_p()
        _check(_lib.gxd_join_build(comm._h, gx_dtype(_np_dtype(self._keys)), self._keys.data_ptr(), self._keys.numel(), int(force_exchange),
                                   _stream(), ctypes.byref(self._h)), "gxd_join_build")

    def inner_join(self, probe_keys: torch.Tensor, chunks: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        probe_keys = probe_keys.contiguous()
        if probe_keys.dtype != self._keys.dtype:
            raise TypeError("Mismatch in joining column data types")
        res = _Results()
        ol, orr, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        try:
            _check(_lib.gxd_join_probe(self._h, probe_keys.data_ptr(), probe_keys.numel(), chunks, res.cb, None, ctypes.byref(ol), ctypes.byref(orr),
                                       ctypes.byref(n), _stream()), "gxd_join_probe")
            return res.take(ol.value, n.value, torch.int64), res.take(orr.value, n.value, torch.int64)
        finally:
            res.release()

    def close(self):
        if self._h:
            _lib.gxd_join_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_ranks(comms: "List[Communicator]", fn):
    """Drive the logical ranks of `Communicator.loopback`: one host thread and one (non-blocking) torch stream per rank, each
    calling fn(rank, comm); returns the per-rank results in rank order, re-raises the first failure.  The sharded operators are
    collective, so the ranks must run concurrently -- ctypes releases the GIL for the duration of every gxd_* call."""
    import threading
    dev = torch.cuda.current_device()
    torch.cuda.synchronize()  # inputs made on the default stream are ready for every rank's stream
    results: List[object] = [None] * len(comms)
    errors: List[Optional[BaseException]] = [None] * len(comms)

    def work(r):
        try:
            torch.cuda.set_device(dev)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                results[r] = fn(r, comms[r])
            s.synchronize()
        except BaseException as e:  # noqa: BLE001 -- reported to the caller below
            errors[r] = e

    threads = [threading.Thread(target=work, args=(r,), name=f"gxd-rank-{r}") for r in range(len(comms))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return results
