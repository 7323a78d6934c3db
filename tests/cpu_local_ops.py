"""TEST-ONLY NumPy implementation of cudf_amd.distributed's LocalOps interface, so the exchange
logic (counts all-to-all, splitters, all_to_all_single with uneven splits, global row ids) can run
under gloo on CPU tensors.  The product never imports this file."""
import numpy as np
import torch

from oracle import cudf_oracle as orc


class NumpyLocalOps:
    def sort(self, keys):
        return torch.from_numpy(np.sort(keys.numpy(), kind="stable"))

    def hash_partition(self, keys, nparts):
        h = orc.murmur3_32(keys.numpy())
        part = (h % np.uint32(nparts)).astype(np.int64)
        gmap = np.argsort(part, kind="stable").astype(np.int32)
        counts = np.bincount(part, minlength=nparts)
        offs = np.concatenate([[0], np.cumsum(counts)])
        return torch.from_numpy(gmap), [int(x) for x in offs]

    def range_partition(self, keys, splitters):
        k = keys.numpy()
        dest = np.searchsorted(np.asarray(splitters, dtype=k.dtype), k, side="right") if len(splitters) else np.zeros(len(k), np.int64)
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=len(splitters) + 1)
        return torch.from_numpy(k[order]), [int(x) for x in np.concatenate([[0], np.cumsum(counts)])]

    def hash_partition_rows(self, keys, nparts):
        gmap, offs = self.hash_partition(keys, nparts)
        return keys[gmap.to(torch.int64)], gmap, offs

    def merge_sum_count(self, keys, sums, counts):
        k, s, c = keys.numpy(), sums.numpy(), counts.numpy()
        uk, inv = np.unique(k, return_inverse=True)
        ms = np.zeros(len(uk), s.dtype)
        mc = np.zeros(len(uk), np.int64)
        np.add.at(ms, inv, s)
        np.add.at(mc, inv, c)
        return torch.from_numpy(uk), torch.from_numpy(ms), torch.from_numpy(mc)

    def gather(self, values, gather_map):
        return values[gather_map.to(torch.int64)]

    def inner_join(self, left, right):
        l, r = orc.inner_join(left.numpy(), right.numpy())
        return torch.from_numpy(l.astype(np.int32)), torch.from_numpy(r.astype(np.int32))

    def gather_global_rows(self, rows, idx, recv, bases):
        starts = np.concatenate([[0], np.cumsum(recv)])
        i = idx.numpy().astype(np.int64)
        seg = np.searchsorted(starts[1:], i, side="right")
        return torch.from_numpy(rows.numpy().astype(np.int64)[i] + np.asarray(bases, np.int64)[seg])

    def join_build(self, right):
        return right.numpy().copy()

    def join_probe(self, table, left):
        l, r = orc.inner_join(left.numpy(), table)
        return torch.from_numpy(l.astype(np.int32)), torch.from_numpy(r.astype(np.int32))

    def groupby_sum_count(self, keys, vals):
        k, res = orc.groupby_agg(keys.numpy(), vals.numpy(), ["sum", "count_valid"], exact=False)
        return torch.from_numpy(k), torch.from_numpy(res["sum"][0]), torch.from_numpy(res["count_valid"][0])

    def reduce(self, values, op):
        v = values.numpy()
        acc = {"sum": np.float64 if v.dtype.kind == "f" else np.int64,
               "product": np.float64 if v.dtype.kind == "f" else np.int64}.get(op, v.dtype)
        if v.size == 0:
            ident = {"sum": 0, "product": 1}.get(op)
            if ident is None:
                lim = np.finfo(v.dtype) if v.dtype.kind == "f" else np.iinfo(v.dtype)
                ident = (np.inf if v.dtype.kind == "f" else lim.max) if op == "min" else (-np.inf if v.dtype.kind == "f" else lim.min)
            return torch.from_numpy(np.array([ident], dtype=acc))
        r, _ = orc.reduce(v, op, None, np.dtype(acc))
        return torch.from_numpy(np.array([r], dtype=acc))

    def scan(self, values, op, inclusive):
        out, _ = orc.scan(values.numpy(), op, inclusive)
        return torch.from_numpy(out)
