"""One-process-per-GPU forms of the hot path: torch.distributed over RCCL on
MI355X (backend "nccl"), gloo in the CPU tests.

Each function takes this rank's SHARD and returns this rank's shard of the result.  The data path
has exactly one exchange step per operation -- an all-to-all over the xGMI mesh (point-to-point
links, so the full mesh keeps all 7 links of a GPU busy; a ring all-reduce would be per-link
bound) -- preceded by an all-gather of a few scalars (counts / splitters).  Reference analogues:
cudf::hash_partition + the rapidsmpf shuffler (cpp/libcudf_streaming/src/partition_utils.cpp:72-117),
cudf_polars' sample -> allgather boundaries -> shuffle -> local sort
(python/cudf_polars/cudf_polars/streaming/actor_graph/collectives/sort.py) and its piecewise
groupby (streaming/groupby.py:411-437).

The local work goes through a `LocalOps` object.  The product implementation (`HipLocalOps`) calls
the HIP kernels through cudf_amd.ops; the CPU tests inject a NumPy implementation of the same
interface (tests/cpu_local_ops.py) to exercise the exchange logic with gloo -- this module itself
never touches the oracle.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------
# local kernels behind an interface
# ------------------------------------------------------------------------------------------------
class HipLocalOps:
    """Local operators on CUDA tensors, implemented by libcudf_amd.so (no CPU path)."""

    def __init__(self):
        from . import ops  # raises if the HIP library is missing
        from .column import Column
        self._ops = ops
        self._Column = Column

    def _col(self, t: torch.Tensor):
        t = t.contiguous()
        return self._Column(t.view(torch.uint8).reshape(-1), np.dtype(str(t.dtype).replace("torch.", "")), t.numel())

    @staticmethod
    def _tensor(col, dtype: torch.dtype) -> torch.Tensor:
        return col.data[: col.size * col.dtype.itemsize].view(dtype)

    def sort(self, keys: torch.Tensor) -> torch.Tensor:
        return self._tensor(self._ops.sort(self._col(keys)), keys.dtype)

    def range_partition(self, keys: torch.Tensor, splitters: Sequence) -> Tuple[torch.Tensor, List[int]]:
        """One pass: keys grouped by destination = number of splitters <= key; (grouped keys, offsets)."""
        if len(splitters) + 1 > self.MAX_PARTS:
            raise ValueError(f"the one-pass exchange partition supports at most {self.MAX_PARTS} ranks (got {len(splitters) + 1})")
        pk, _, offs = self._ops.partition_rows(self._col(keys), len(splitters) + 1, splitters=list(splitters), want_rows=False)
        return self._tensor(pk, keys.dtype), offs

    MAX_PARTS = 16  # gx_partition_rows splits into <= 16 groups in one pass; gx_gather_global_rows resolves <= 16 segments

    def hash_partition_rows(self, keys: torch.Tensor, nparts: int) -> Tuple[torch.Tensor, torch.Tensor, List[int]]:
        """One pass: (keys grouped by destination rank, their int32 local rows, offsets).  The destination is a hash
        that is independent of the local join table's slot bits."""
        if nparts > self.MAX_PARTS:
            raise ValueError(f"the one-pass exchange partition supports at most {self.MAX_PARTS} ranks (got {nparts}); "
                             "run a two-level exchange (node x GPU) above it")
        if nparts & (nparts - 1):  # not a power of two ranks: murmur3 % nparts through the partition map
            m, offs = self._ops.hash_partition_map([self._col(keys)], nparts)
            gm = self._tensor(m, torch.int32)
            return self.gather(keys, gm), gm, [int(x) for x in offs]
        pk, rows, offs = self._ops.partition_rows(self._col(keys), nparts)
        return self._tensor(pk, keys.dtype), self._tensor(rows, torch.int32), offs

    def gather(self, values: torch.Tensor, gather_map: torch.Tensor) -> torch.Tensor:
        out = self._ops.gather(self._col(values), self._col(gather_map))
        return self._tensor(out, values.dtype)

    def inner_join(self, left: torch.Tensor, right: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        li, ri = self._ops.inner_join(self._col(left), self._col(right))
        return self._tensor(li, torch.int32), self._tensor(ri, torch.int32)

    def gather_global_rows(self, rows: torch.Tensor, idx: torch.Tensor, recv: Sequence[int], bases: Sequence[int]) -> torch.Tensor:
        """int64 global row of every selected entry: rows[idx] + the shard offset of the rank its segment came from"""
        out = self._ops.gather_global_rows(self._col(rows), self._col(idx), recv, bases)
        return self._tensor(out, torch.int64)

    def join_build(self, right: torch.Tensor):
        """cudf::hash_join on the local build keys: hashed once, probed many times"""
        col = self._col(right)
        return (self._ops.HashJoin(col), col)

    def join_probe(self, table, left: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        li, ri = table[0].inner_join(self._col(left))
        return self._tensor(li, torch.int32), self._tensor(ri, torch.int32)

    def groupby_sum_count(self, keys: torch.Tensor, vals: torch.Tensor):
        k, s, cv, _ = self._ops.groupby_sum_count(self._col(keys), self._col(vals))
        sdt = vals.dtype if vals.dtype.is_floating_point else torch.int64
        return self._tensor(k, keys.dtype), self._tensor(s, sdt), self._tensor(cv, torch.int32)

    def merge_sum_count(self, keys: torch.Tensor, sums: torch.Tensor, counts: torch.Tensor):
        """Partial (key, sum, count) rows -> one row per key, keys ascending: ONE grouping (sort the keys, run
        boundaries) shared by the two segmented reductions."""
        k, s, c = self._ops.merge_sum_count(self._col(keys), self._col(sums), self._col(counts))
        return self._tensor(k, keys.dtype), self._tensor(s, sums.dtype), self._tensor(c, torch.int64)

    def reduce(self, values: torch.Tensor, op: str) -> torch.Tensor:
        """1-element tensor: cudf::reduce of the shard in the accumulator type (int64 / float64 for
        sum and product, the input type for min / max); the operator's identity for an empty shard."""
        acc = _acc_dtype(values.dtype, op)
        if values.numel() == 0:
            return torch.tensor([_identity(acc, op)], dtype=acc, device=values.device)
        v, _ = self._ops.reduce(self._col(values), op, out_dtype=np.dtype(str(acc).replace("torch.", "")))
        return torch.tensor([v.item()], dtype=acc, device=values.device)

    def scan(self, values: torch.Tensor, op: str, inclusive: bool) -> torch.Tensor:
        return self._tensor(self._ops.scan(self._col(values), op, inclusive), values.dtype)


def _acc_dtype(dtype: torch.dtype, op: str) -> torch.dtype:
    if op in ("sum", "product"):
        return torch.float64 if dtype.is_floating_point else torch.int64
    return dtype


def _identity(dtype: torch.dtype, op: str):
    if op == "sum":
        return 0
    if op == "product":
        return 1
    if dtype.is_floating_point:
        return float("inf") if op == "min" else float("-inf")
    info = torch.iinfo(dtype)
    return info.max if op == "min" else info.min


def _combine(a: torch.Tensor, b: torch.Tensor, op: str) -> torch.Tensor:
    if op == "sum":
        return a + b
    if op == "product":
        return a * b
    return torch.minimum(a, b) if op == "min" else torch.maximum(a, b)


# ------------------------------------------------------------------------------------------------
# exchange primitives
# ------------------------------------------------------------------------------------------------
def _world(group) -> Tuple[int, int]:
    return dist.get_rank(group), dist.get_world_size(group)


def exchange_counts(send_counts: Sequence[int], device, group=None) -> List[int]:
    """all-to-all of one int64 per peer: recv_counts[j] = what rank j sends to me."""
    _, world = _world(group)
    s = torch.tensor(list(send_counts), dtype=torch.int64, device=device)
    r = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(r, s, group=group)
    return [int(x) for x in r.cpu()]


def all_to_all_rows(data: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int], group=None) -> torch.Tensor:
    """Rows [sum(send_counts[:j]), +send_counts[j]) of `data` go to rank j; returns the rows received,
    ordered by source rank.  One collective on the full xGMI mesh."""
    out = torch.empty(int(sum(recv_counts)), dtype=data.dtype, device=data.device)
    dist.all_to_all_single(out, data.contiguous(), output_split_sizes=list(recv_counts),
                           input_split_sizes=list(send_counts), group=group)
    return out


def _offsets_to_counts(offsets: Sequence[int]) -> List[int]:
    return [int(offsets[i + 1] - offsets[i]) for i in range(len(offsets) - 1)]


# ------------------------------------------------------------------------------------------------
# distributed operators
# ------------------------------------------------------------------------------------------------
_FORCE_EXCHANGE = False  # tests: take the partition + all-to-all path even when world == 1

# The PRODUCT path on GPUs is C++ over RCCL (include/cudf_amd/gxd.h, cudf_amd/cpp/src/distributed.cpp): chunked partition
# passes with the exchange of chunk k overlapping the partition of chunk k + 1 and (join) the probe of chunk k - 1, counts by
# ncclAllGather, rows by grouped ncclSend / ncclRecv, persistent buffers.  The functions below hand CUDA tensors to it whenever no
# LocalOps object is injected; the torch.distributed implementation that follows is what the gloo tests drive with a NumPy
# LocalOps, and it documents the exchange logic in 60 lines.
_COMMS = {}


def _gxd_comm(group):
    from . import gxd
    key = id(group) if group is not None else 0
    if key not in _COMMS:
        _COMMS[key] = gxd.Communicator(group)
    return _COMMS[key]


def close_communicators():
    """destroy the RCCL communicators this module created (before dist.destroy_process_group)"""
    for c in _COMMS.values():
        c.close()
    _COMMS.clear()


def abort_communicators():
    """from a watchdog thread: make a sharded operator that is blocked inside a collective return an error (gxd_comm_abort ->
    ncclCommAbort).  The communicators are unusable afterwards; close_communicators still releases their buffers."""
    for c in list(_COMMS.values()):
        c.abort()


def _use_gxd(local, t: torch.Tensor) -> bool:
    return local is None and t.is_cuda


def distributed_sort(keys: torch.Tensor, local: Optional[object] = None, group=None, samples_per_rank: int = 1024) -> torch.Tensor:
    """Global sort of the concatenation of all ranks' shards; rank r returns the r-th range, so the
    concatenation of the results in rank order is sorted.  Sample sort with ONE partition pass and ONE sort per
    rank: strided sample of the UNSORTED shard -> all-gather -> common splitters -> one range-partition pass
    into `world` send regions -> all-to-all -> local sort of what arrived
    (cudf_polars: sample -> allgather boundaries -> shuffle -> local sort, collectives/sort.py)."""
    if _use_gxd(local, keys):
        return _gxd_comm(group).sort(keys, force_exchange=_FORCE_EXCHANGE)
    local = local or HipLocalOps()
    rank, world = _world(group)
    if world == 1 and not _FORCE_EXCHANGE:
        return local.sort(keys)
    n = keys.numel()
    # evenly strided sample of the shard as it is (empty shards contribute the dtype's max so they never split)
    if n > 0:
        # integer arithmetic: float32 cannot represent n - 1 for n ~ 1e9 (it would round to n)
        pos = (torch.arange(samples_per_rank, dtype=torch.int64, device=keys.device) * (n - 1)) // max(samples_per_rank - 1, 1)
        mine = keys[pos]
    else:
        fill = torch.finfo(keys.dtype).max if keys.dtype.is_floating_point else torch.iinfo(keys.dtype).max
        mine = torch.full((samples_per_rank,), fill, dtype=keys.dtype, device=keys.device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    allsamp = local.sort(torch.cat(gathered))
    cut = torch.arange(1, world, device=keys.device) * samples_per_rank
    splitters = allsamp[cut].cpu().tolist()                   # world-1 values, identical on every rank
    pk, offs = local.range_partition(keys, splitters)          # rank j gets keys with j splitters <= key
    send = _offsets_to_counts(offs)
    recv = exchange_counts(send, keys.device, group)
    got = all_to_all_rows(pk, send, recv, group)
    return local.sort(got)


def _all_sizes(n: int, device, group) -> List[int]:
    _, world = _world(group)
    sizes = [torch.empty(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=device), group=group)
    return [int(x) for x in sizes]


def distributed_inner_join(left: torch.Tensor, right: torch.Tensor, local: Optional[object] = None, group=None
                           ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inner equi-join of the concatenations of all ranks' `left` and `right` key shards.
    Returns this rank's share of the result as (global_left_row, global_right_row) int64 tensors,
    where a global row id = (offset of the owning rank's shard) + local row.  Both sides are hash-partitioned in
    one pass each, exchanged once as (key, int32 local row) = 12 B/row -- the source rank, hence the shard offset, is
    implied by the receive segment -- and joined locally; outputs stay sharded.  The build side's exchange is in
    flight while the probe side is partitioned."""
    if _use_gxd(local, left):
        from . import gxd
        hj = gxd.HashJoin(_gxd_comm(group), right, force_exchange=_FORCE_EXCHANGE)
        try:
            return hj.inner_join(left)
        finally:
            hj.close()
    local = local or HipLocalOps()
    rank, world = _world(group)
    dev = left.device
    lsizes, rsizes = _all_sizes(left.numel(), dev, group), _all_sizes(right.numel(), dev, group)

    def bases(sizes):
        out, run = [], 0
        for x in sizes:
            out.append(run)
            run += x
        return out

    if world == 1 and not _FORCE_EXCHANGE:
        li, ri = local.inner_join(left, right)
        return li.to(torch.int64), ri.to(torch.int64)

    def start_exchange(pk, prow, offs):
        send = _offsets_to_counts(offs)
        recv = exchange_counts(send, dev, group)
        ok = torch.empty(int(sum(recv)), dtype=pk.dtype, device=dev)
        orow = torch.empty(int(sum(recv)), dtype=torch.int32, device=dev)
        w1 = dist.all_to_all_single(ok, pk.contiguous(), output_split_sizes=list(recv), input_split_sizes=list(send), group=group, async_op=True)
        w2 = dist.all_to_all_single(orow, prow.contiguous(), output_split_sizes=list(recv), input_split_sizes=list(send), group=group, async_op=True)
        return ok, orow, recv, (w1, w2)

    rk, rrow, roffs = local.hash_partition_rows(right, world)
    rkeys, rrows, rrecv, rwork = start_exchange(rk, rrow, roffs)     # build side in flight ...
    lk, lrow, loffs = local.hash_partition_rows(left, world)        # ... while the probe side is partitioned
    lkeys, lrows, lrecv, lwork = start_exchange(lk, lrow, loffs)
    for w in rwork + lwork:
        w.wait()
    li, ri = local.inner_join(lkeys, rkeys)
    return (local.gather_global_rows(lrows, li, lrecv, bases(lsizes)),
            local.gather_global_rows(rrows, ri, rrecv, bases(rsizes)))


class DistributedHashJoin:
    """cudf::hash_join over sharded tables (hash_join.hpp:70-125: build once, probe many).  The constructor hash-partitions
    this rank's build keys, exchanges them once (key + int32 local row, 12 B/row) and builds the local hash table over
    what it received; inner_join() partitions and exchanges only the probe side and probes that table.  Results as in
    distributed_inner_join: (global probe row, global build row) of this rank's share of the pairs."""

    def __init__(self, right: torch.Tensor, local: Optional[object] = None, group=None):
        self._gxd = None
        if _use_gxd(local, right):
            from . import gxd
            self._gxd = gxd.HashJoin(_gxd_comm(group), right, force_exchange=_FORCE_EXCHANGE)
            return
        self._local = local or HipLocalOps()
        self._group = group
        rank, world = _world(group)
        dev = right.device
        rsizes = _all_sizes(right.numel(), dev, group)
        self._single = world == 1 and not _FORCE_EXCHANGE
        if self._single:
            self._table = self._local.join_build(right)
            return
        rk, rrow, roffs = self._local.hash_partition_rows(right, world)
        send = _offsets_to_counts(roffs)
        recv = exchange_counts(send, dev, group)
        rkeys = all_to_all_rows(rk, send, recv, group)
        self._rrows, self._rrecv, self._rbases = all_to_all_rows(rrow, send, recv, group), list(recv), _bases(rsizes)
        self._table = self._local.join_build(rkeys)

    def inner_join(self, left: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._gxd is not None:
            return self._gxd.inner_join(left)
        local, group = self._local, self._group
        if self._single:
            li, ri = local.join_probe(self._table, left)
            return li.to(torch.int64), ri.to(torch.int64)
        rank, world = _world(group)
        dev = left.device
        lsizes = _all_sizes(left.numel(), dev, group)
        lk, lrow, loffs = local.hash_partition_rows(left, world)
        send = _offsets_to_counts(loffs)
        recv = exchange_counts(send, dev, group)
        lkeys = all_to_all_rows(lk, send, recv, group)
        lrows = all_to_all_rows(lrow, send, recv, group)
        li, ri = local.join_probe(self._table, lkeys)
        # only the rows that joined are translated: (int32 local row, segment) -> int64 global row in one gather each
        return (local.gather_global_rows(lrows, li, recv, _bases(lsizes)),
                local.gather_global_rows(self._rrows, ri, self._rrecv, self._rbases))


def _bases(sizes: Sequence[int]) -> List[int]:
    out, run = [], 0
    for x in sizes:
        out.append(run)
        run += x
    return out


def distributed_groupby_sum_count(keys: torch.Tensor, vals: torch.Tensor, local: Optional[object] = None, group=None):
    """groupby(keys).agg(sum, count) over all ranks' shards.  Pre-aggregate locally (<= #groups rows),
    hash-partition the partials, one all-to-all, merge.  Every group ends on exactly one rank.
    Returns (keys, sum, count) for the groups this rank owns (count as int64), keys ascending after a merge."""
    if _use_gxd(local, keys):
        return _gxd_comm(group).groupby_sum_count(keys, vals, force_exchange=_FORCE_EXCHANGE)
    local = local or HipLocalOps()
    rank, world = _world(group)
    k, s, c = local.groupby_sum_count(keys, vals)
    c = c.to(torch.int64)
    if world == 1 and not _FORCE_EXCHANGE:
        return k, s, c
    pk, prow, offs = local.hash_partition_rows(k, world)
    send = _offsets_to_counts(offs)
    recv = exchange_counts(send, k.device, group)
    rk = all_to_all_rows(pk, send, recv, group)
    rs = all_to_all_rows(local.gather(s, prow), send, recv, group)
    rc = all_to_all_rows(local.gather(c, prow), send, recv, group)
    # one grouping of the received partials carries both the sum and the count
    return local.merge_sum_count(rk, rs, rc)


def _gather_partials(part: torch.Tensor, group=None) -> List[torch.Tensor]:
    _, world = _world(group)
    parts = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(parts, part, group=group)
    return parts


def distributed_reduce(values: torch.Tensor, op: str = "sum", local: Optional[object] = None, group=None):
    """cudf::reduce over the concatenation of all ranks' shards (an all-gather of one
    partial per GPU).  Every rank returns the same Python scalar; partials are folded in rank order, so
    the floating-point result does not depend on which rank asks."""
    local = local or HipLocalOps()
    parts = _gather_partials(local.reduce(values, op), group)
    acc = parts[0]
    for p in parts[1:]:
        acc = _combine(acc, p, op)
    return acc.item()


def distributed_scan(values: torch.Tensor, op: str = "sum", inclusive: bool = True, local: Optional[object] = None,
                     group=None) -> torch.Tensor:
    """cudf::scan over the concatenation of all ranks' shards in rank order; rank r returns its shard of
    the result (same dtype as the input: integers wrap like the single-GPU scan).  One all-gather of the
    shard totals; the exclusive prefix of the totals of the ranks before this one is folded into the
    first element of a copy of the shard, so the local scan kernel produces the global values in its one pass."""
    local = local or HipLocalOps()
    rank, _ = _world(group)
    dt = values.dtype
    total = local.reduce(values, op)
    if op in ("sum", "product") and not dt.is_floating_point:
        total = total.to(dt)  # wrap to the scan's element type: (a + b) mod 2^w is associative
    elif dt.is_floating_point:
        total = total.to(dt)
    parts = _gather_partials(total, group)
    if rank == 0 or values.numel() == 0:
        return local.scan(values, op, inclusive)
    prefix = parts[0]
    for p in parts[1:rank]:
        prefix = _combine(prefix, p, op)
    # the prefix of the earlier ranks is folded into a COPY of the shard's first element: the caller's tensor is never
    # written (another stream may be reading it)
    patched = values.clone()
    patched[:1] = _combine(prefix, values[:1], op)
    out = local.scan(patched, op, inclusive)
    if not inclusive:
        out[:1] = prefix  # an exclusive scan starts at the identity: here at everything before this shard
    return out
