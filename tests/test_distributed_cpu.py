"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups drive cudf_amd.distributed's exchange
logic (all-to-all with uneven splits, splitter agreement, global row ids) with a NumPy LocalOps
(tests/cpu_local_ops.py).  Results are compared with the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shards(rank, world, what):
    """deterministic ragged shards (rank 1 of the sort case is empty on purpose)"""
    rng = np.random.default_rng(100 + rank)
    if what == "sort":
        n = [5000, 0, 12345][rank % 3]
        return rng.integers(-2**62, 2**62, n, dtype=np.int64)
    if what == "join":
        left = rng.integers(0, 3000, 4000 + 500 * rank).astype(np.int64)
        right = (rng.permutation(4000)[: 1500 + 100 * rank]).astype(np.int64)
        return left, right
    if what.startswith("scan") or what.startswith("reduce"):
        n = [4001, 0, 2500][rank % 3]                           # rank 1 is empty on purpose
        dt = what.split(":")[2]
        if dt == "int32":
            return rng.integers(-2**31, 2**31, n).astype(np.int32)   # wraps
        return rng.integers(-1000, 1000, n).astype(dt)            # exact in float64
    keys = rng.integers(0, 700, 20000 + 1000 * rank).astype(np.int32)
    vals = rng.integers(0, 100, len(keys)).astype(np.float64)  # small integers: sums are exact in any order
    return keys, vals


def _worker(rank, world, port, what, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("cudf_amd_distributed", os.path.join(ROOT, "cudf_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)  # load the module file directly: importing the package needs the HIP library
    spec.loader.exec_module(D)
    from tests.cpu_local_ops import NumpyLocalOps
    local = NumpyLocalOps()
    if what == "sort":
        out = D.distributed_sort(torch.from_numpy(_shards(rank, world, what)), local=local, samples_per_rank=16)
        np.save(os.path.join(outdir, f"sort_{rank}.npy"), out.numpy())
    elif what == "join":
        l, r = _shards(rank, world, what)
        gl, gr = D.distributed_inner_join(torch.from_numpy(l), torch.from_numpy(r), local=local)
        np.save(os.path.join(outdir, f"join_{rank}.npy"), np.stack([gl.numpy(), gr.numpy()]))
    elif what == "hashjoin":   # build exchanged and hashed once, probed twice (the second probe: the first half of the shard)
        l, r = _shards(rank, world, "join")
        hj = D.DistributedHashJoin(torch.from_numpy(r), local=local)
        gl, gr = hj.inner_join(torch.from_numpy(l))
        np.save(os.path.join(outdir, f"hashjoin_{rank}.npy"), np.stack([gl.numpy(), gr.numpy()]))
        gl2, gr2 = hj.inner_join(torch.from_numpy(l[: len(l) // 2].copy()))
        np.save(os.path.join(outdir, f"hashjoin2_{rank}.npy"), np.stack([gl2.numpy(), gr2.numpy()]))
    elif what.startswith("scan"):
        _, op, dt, inc = what.split(":")
        v = _shards(rank, world, what)
        keep = v.copy()
        out = D.distributed_scan(torch.from_numpy(v), op, inc == "inc", local=local)
        assert np.array_equal(v, keep)                          # the input is restored
        np.save(os.path.join(outdir, f"scan_{rank}.npy"), out.numpy())
    elif what.startswith("reduce"):
        _, op, dt = what.split(":")
        r = D.distributed_reduce(torch.from_numpy(_shards(rank, world, what)), op, local=local)
        np.save(os.path.join(outdir, f"reduce_{rank}.npy"), np.array([r]))
    else:
        k, v = _shards(rank, world, what)
        gk, gs, gc = D.distributed_groupby_sum_count(torch.from_numpy(k), torch.from_numpy(v), local=local)
        np.save(os.path.join(outdir, f"gb_{rank}.npy"), np.stack([gk.numpy().astype(np.float64), gs.numpy(), gc.numpy().astype(np.float64)]))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, what, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), what, str(tmp_path)), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_sort_gloo(world, tmp_path):
    _run(world, "sort", tmp_path)
    parts = [np.load(tmp_path / f"sort_{r}.npy") for r in range(world)]
    got = np.concatenate(parts)
    exp = np.sort(np.concatenate([_shards(r, world, "sort") for r in range(world)]))
    np.testing.assert_array_equal(got, exp)  # rank order == global order
    for a, b in zip(parts[:-1], parts[1:]):
        if len(a) and len(b):
            assert a[-1] <= b[0]


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_hash_join_builds_once_probes_many_gloo(world, tmp_path):
    from oracle import cudf_oracle as orc
    _run(world, "hashjoin", tmp_path)
    shards = [_shards(r, world, "join") for r in range(world)]
    left = np.concatenate([s[0] for s in shards])
    right = np.concatenate([s[1] for s in shards])
    pairs = np.concatenate([np.load(tmp_path / f"hashjoin_{r}.npy") for r in range(world)], axis=1)
    el, er = orc.inner_join(left, right)
    gl, gr = orc.canonical_pairs(pairs[0], pairs[1])
    np.testing.assert_array_equal(gl, el)
    np.testing.assert_array_equal(gr, er)
    # second probe: every rank's first half; global probe ids count positions in the concatenation of the halves
    halves = [s[0][: len(s[0]) // 2] for s in shards]
    pairs2 = np.concatenate([np.load(tmp_path / f"hashjoin2_{r}.npy") for r in range(world)], axis=1)
    el2, er2 = orc.inner_join(np.concatenate(halves), right)
    gl2, gr2 = orc.canonical_pairs(pairs2[0], pairs2[1])
    np.testing.assert_array_equal(gl2, el2)
    np.testing.assert_array_equal(gr2, er2)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_inner_join_gloo(world, tmp_path):
    from oracle import cudf_oracle as orc
    _run(world, "join", tmp_path)
    pairs = np.concatenate([np.load(tmp_path / f"join_{r}.npy") for r in range(world)], axis=1)
    shards = [_shards(r, world, "join") for r in range(world)]
    left = np.concatenate([s[0] for s in shards])
    right = np.concatenate([s[1] for s in shards])
    el, er = orc.inner_join(left, right)
    gl, gr = orc.canonical_pairs(pairs[0], pairs[1])
    np.testing.assert_array_equal(gl, el)   # global row ids = position in the rank-order concatenation
    np.testing.assert_array_equal(gr, er)


@pytest.mark.parametrize("world", [2])
def test_distributed_groupby_gloo(world, tmp_path):
    _run(world, "groupby", tmp_path)
    res = np.concatenate([np.load(tmp_path / f"gb_{r}.npy") for r in range(world)], axis=1)
    shards = [_shards(r, world, "groupby") for r in range(world)]
    keys = np.concatenate([s[0] for s in shards])
    vals = np.concatenate([s[1] for s in shards])
    o = np.argsort(res[0])
    uk = np.unique(keys)
    np.testing.assert_array_equal(res[0][o], uk.astype(np.float64))      # every group exactly once
    np.testing.assert_array_equal(res[1][o], np.bincount(keys, weights=vals)[uk])
    np.testing.assert_array_equal(res[2][o], np.bincount(keys)[uk].astype(np.float64))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("op,dt,inc", [("sum", "int64", "inc"), ("sum", "int32", "exc"), ("sum", "float64", "inc"),
                                       ("min", "int64", "inc"), ("max", "float64", "exc")])
def test_distributed_scan_gloo(world, op, dt, inc, tmp_path):
    from oracle import cudf_oracle as orc
    what = f"scan:{op}:{dt}:{inc}"
    _run(world, what, tmp_path)
    got = np.concatenate([np.load(tmp_path / f"scan_{r}.npy") for r in range(world)])
    full = np.concatenate([_shards(r, world, what) for r in range(world)])
    exp, _ = orc.scan(full, op, inc == "inc")
    assert got.dtype == np.dtype(dt)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("op,dt", [("sum", "int64"), ("sum", "float64"), ("min", "int64"), ("max", "float64")])
def test_distributed_reduce_gloo(world, op, dt, tmp_path):
    from oracle import cudf_oracle as orc
    what = f"reduce:{op}:{dt}"
    _run(world, what, tmp_path)
    res = [np.load(tmp_path / f"reduce_{r}.npy")[0] for r in range(world)]
    full = np.concatenate([_shards(r, world, what) for r in range(world)])
    exp, _ = orc.reduce(full, op, None, np.float64 if (dt == "float64" and op == "sum") else (np.int64 if op == "sum" else None))
    assert all(r == res[0] for r in res)                       # every rank returns the same scalar
    assert res[0] == exp
