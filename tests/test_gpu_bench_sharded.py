"""GPU: what `bench.py --gpus N` executes before and around its timed steps -- the oracle-checked PRE-FLIGHT of the shipped
sharded operators under a watchdog, the full-size guards of the sharded lines -- run here (a) over RCCL with TWO ranks when the
box has two devices (`RcclTransport` with W > 1; uneven shards; from 3 ranks on one shard is empty), and
(b) in a 1-rank group with the exchange forced (`--force-sharded`) so that every line of that code path runs on the 1-GPU boxes
the test tier uses.  Reference analogue of what is exercised: the shuffle of cpp/libcudf_streaming/src/partition_utils.cpp:72-117
and partition.cpp:56-80 (count all-gather, grouped send / receive, local operator).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.pop("LOCAL_RANK", None)
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def _assert_preflight_ok(pf, ranks):
    assert pf["ranks"] == ranks
    assert pf["communicator_alive"] is True
    for op in ("sort_fused", "sort_sample", "join", "groupby"):
        assert str(pf[op]).startswith("ok"), (op, pf[op])


def test_preflight_single_rank_forced_exchange():
    """the pre-flight end to end on one device: gxd_sort (fused + sample-sort path), gxd_join_build / probe, gxd_groupby_sum_count
    with the exchange forced, gathered to rank 0 and compared with the oracle; exit code 0 and an `ok` verdict per operator"""
    r, line = _run(["--force-sharded", "--preflight-only"], 600)
    assert r.returncode == 0, r.stderr[-3000:]
    _assert_preflight_ok(line["preflight"], 1)
    assert "fused path" in line["preflight"]["sort_fused"]


def test_sharded_lines_single_rank_forced_exchange():
    """the `--gpus N` line itself (sort head + join + groupby blocks) in a 1-rank group: the full-size guards -- shard order over
    the ranks, pair count, the hit rows, EVERY pair's keys fetched from the owners of its global rows, group totals -- execute and
    pass, and the line carries the pre-flight verdict"""
    r, line = _run(["--force-sharded", "--rows", "40000000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], 900)
    assert r.returncode == 0, r.stderr[-3000:]
    _assert_preflight_ok(line["preflight"], 1)
    assert "C++ operators over RCCL" in line["config"]["workload"]
    assert "every pair joins equal keys" in line["join"]["checked"]
    assert "C++ operators over RCCL" in line["join"]["config"]["workload"]
    assert "C++ operators over RCCL" in line["groupby"]["config"]["workload"]
    assert line["value"] > 0 and line["join"]["value"] > 0 and line["groupby"]["value"] > 0


def test_preflight_two_ranks_over_rccl():
    """TWO processes, two devices, the product transport: ncclCommInitRank with a broadcast id, ncclAllGather of the count rows,
    grouped ncclSend / ncclRecv of uneven spans -- the shipped operators against the oracle on the concatenated input"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU box)")
    r, line = _run(["--gpus", "2", "--preflight-only"], 900)
    assert r.returncode == 0, r.stderr[-3000:]
    _assert_preflight_ok(line["preflight"], 2)
    assert "RCCL" in line["preflight"]["transport"]


def test_sharded_lines_two_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU box)")
    r, line = _run(["--gpus", "2", "--rows", "40000000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], 1200)
    assert r.returncode == 0, r.stderr[-3000:]
    _assert_preflight_ok(line["preflight"], 2)
    assert line["n_gpus"] == 2
    assert "every pair joins equal keys" in line["join"]["checked"]
