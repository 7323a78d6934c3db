"""Device column = Arrow-layout buffers in HBM (data + optional validity bitmap).

Mirrors the data model of cudf::column / cudf::column_view
(cpp/include/cudf/column/column.hpp:36-331, column_view.hpp): a typed data buffer, an optional
LSB-first validity bitmap in uint32 words padded to 64 bytes (cpp/include/cudf/null_mask.hpp:55)
and a host-side null count.  PyTorch is used only as the device allocator / stream provider.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib as L

_NP2GX = {
    np.dtype("int8"): L.INT8, np.dtype("int16"): L.INT16, np.dtype("int32"): L.INT32,
    np.dtype("int64"): L.INT64, np.dtype("uint8"): L.UINT8, np.dtype("uint16"): L.UINT16,
    np.dtype("uint32"): L.UINT32, np.dtype("uint64"): L.UINT64, np.dtype("float32"): L.FLOAT32,
    np.dtype("float64"): L.FLOAT64, np.dtype("bool"): L.BOOL8,
}


def gx_dtype(dt) -> int:
    try:
        return _NP2GX[np.dtype(dt)]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}") from None  # cudf::data_type_error -> TypeError


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def device_bytes(nbytes: int) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device="cuda")


def ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def bitmask_words(nbits: int) -> int:
    """uint32 words of a validity bitmap, allocation padded to 64 B (null_mask.hpp:55)."""
    return ((nbits + 511) // 512) * 16


def pack_mask(valid: np.ndarray) -> np.ndarray:
    """bool array -> LSB-first uint32 words (padded)."""
    n = len(valid)
    bits = np.zeros(bitmask_words(n) * 32, dtype=np.uint8)
    bits[:n] = np.asarray(valid, dtype=np.uint8)
    return np.packbits(bits, bitorder="little").view(np.uint32)


def unpack_mask(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


class Column:
    """Owning device column."""

    __slots__ = ("data", "dtype", "size", "mask", "null_count")

    def __init__(self, data: torch.Tensor, dtype, size: int, mask: Optional[torch.Tensor] = None,
                 null_count: int = 0):
        self.data = data          # uint8 tensor, size*itemsize bytes (at least)
        self.dtype = np.dtype(dtype)
        self.size = int(size)
        self.mask = mask          # int32 tensor of bitmask words or None
        self.null_count = int(null_count) if mask is not None else 0

    # -------- construction / export (host <-> device, Arrow layout)
    @classmethod
    def from_numpy(cls, values: np.ndarray, valid: Optional[np.ndarray] = None) -> "Column":
        v = np.ascontiguousarray(values)
        if v.size > 2**31 - 1:
            raise OverflowError("column size exceeds cudf::size_type")  # column.hpp:78-81
        raw = torch.from_numpy(v.view(np.uint8).reshape(-1).copy()) if v.size else torch.empty(0, dtype=torch.uint8)
        data = device_bytes(v.nbytes)
        if v.size:
            data[: v.nbytes].copy_(raw)
        mask = None
        nulls = 0
        if valid is not None:
            valid = np.asarray(valid, dtype=bool)
            nulls = int((~valid).sum())
            words = pack_mask(valid)
            mask = torch.from_numpy(words.view(np.int32).copy()).cuda()
        return cls(data, v.dtype, v.size, mask, nulls)

    @classmethod
    def empty(cls, dtype, size: int, nullable: bool = False) -> "Column":
        dt = np.dtype(dtype)
        data = device_bytes(size * dt.itemsize)
        mask = torch.zeros(bitmask_words(size), dtype=torch.int32, device="cuda") if nullable else None
        return cls(data, dt, size, mask, 0)

    def to_numpy(self) -> np.ndarray:
        nbytes = self.size * self.dtype.itemsize
        return self.data[:nbytes].cpu().numpy().view(self.dtype).copy()

    def valid_numpy(self) -> Optional[np.ndarray]:
        if self.mask is None:
            return None
        return unpack_mask(self.mask.cpu().numpy().view(np.uint32), self.size)

    # -------- raw pointers for the C ABI
    @property
    def gx(self) -> int:
        return gx_dtype(self.dtype)

    @property
    def data_ptr(self):
        return ctypes.c_void_p(self.data.data_ptr())

    @property
    def mask_ptr(self):
        return ctypes.c_void_p(self.mask.data_ptr()) if self.mask is not None else None

    def has_nulls(self) -> bool:
        return self.mask is not None and self.null_count > 0

    def __len__(self):
        return self.size
