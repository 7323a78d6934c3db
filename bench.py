#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: rows/s + achieved HBM GB/s of the hot path.

  python bench.py --gpus N --steps K --warmup W [--workload sort|sorted_order|join|groupby]
                  [--rows R] [--cpu-baseline/--no-cpu-baseline]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM
(generated on the device by a counter-based RNG, so no PCIe traffic inside or outside the timed
region).  Default workload = BASELINE.json configs[1]: 1e9-row int64 radix sort (cudf::sort) on
one GPU.  N > 1: one process per GPU (torch.distributed / RCCL over xGMI), every rank holds a shard
of `--rows` rows and the step is the DISTRIBUTED operator of cudf_amd/distributed.py (sample sort /
hash-partitioned join / pre-aggregated groupby: one all-to-all exchange each, no all-reduce) ->
"scaling": "weak", value = rows of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (the radix scatter pass): algorithmic bytes per launch
                  (16 B/row = read 8 + write 8) / average launch duration measured
                  live with HIP events on the launch stream (gx_sort_profile); peak = 8.0 TB/s HBM3E.
  cpu_baseline -- the CPU oracle ("port": oracle/oracle.c LSD radix sort, 1 thread) timed on the
                  host on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sort", choices=["sort", "sorted_order", "join", "groupby", "reduce", "scan", "gather"])
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--algo", type=int, default=0,
                    help="sort knob: 0 onesweep/windowed look-back, 1 three-kernel, 2 onesweep/one-tile look-back")
    ap.add_argument("--gb-algo", type=int, default=0, help="groupby knob: 0 auto, 1 global table, 2 LDS-partitioned")
    ap.add_argument("--gb-split", type=int, default=1)
    ap.add_argument("--no-partitioned-join", action="store_true", help="join: probe the table directly")
    ap.add_argument("--no-hybrid", action="store_true", help="sort: disable the hybrid MSD path (LSD passes only)")
    ap.add_argument("--key-range", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="sort: keys uniform in [LO, HI) instead of the full int64 range (the reference's own "
                         "benchmark distribution is 100 10001: benchmarks/sort/sort.cpp:24-26)")
    ap.add_argument("--cpu-baseline", dest="cpu", action="store_true", default=True)
    ap.add_argument("--no-cpu-baseline", dest="cpu", action="store_false")
    ap.add_argument("--cpu-rows", type=float, default=0, help="rows of the CPU-baseline sample (0 = per-workload default)")
    return ap.parse_args()


def cpu_baseline_sort(rows):
    """oracle ("port") timed on the host: single-thread LSD radix sort in C + pandas for context."""
    import numpy as np
    from oracle import c_oracle
    n = int(rows)
    rng = np.random.default_rng(42)
    v = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    t0 = time.perf_counter()
    out = c_oracle.sort_i64(v)
    dt = time.perf_counter() - t0
    assert out[0] <= out[n // 2] <= out[-1]
    extra = {}
    try:
        import pandas as pd
        m = min(n, 10_000_000)
        df = pd.DataFrame({"a": v[:m]})
        t0 = time.perf_counter()
        df.sort_values("a", kind="stable")
        extra["pandas_sort_values_rows_per_s"] = m / (time.perf_counter() - t0)
        extra["pandas_rows"] = m
        import pyarrow as pa
        import pyarrow.compute as pc
        arr = pa.array(v[:m])
        t0 = time.perf_counter()
        pc.sort_indices(arr)
        extra["pyarrow_sort_indices_rows_per_s"] = m / (time.perf_counter() - t0)
        extra["pyarrow_threads"] = pa.cpu_count()
    except Exception as e:  # pandas is context only
        extra["pandas_error"] = repr(e)
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"{n} uniform int64 rows (seed 42), oracle/oracle.c orc_sort_i64 (8-pass LSD radix, 1 thread)",
            "host_cpus": os.cpu_count(), **extra}


def cpu_baseline_join(rows):
    import numpy as np
    from oracle import c_oracle
    n = int(rows)
    rng = np.random.default_rng(12345)
    build = rng.permutation(n // 5).astype(np.int64)[: n // 10]
    probe = rng.integers(0, n // 3, n).astype(np.int64)
    t0 = time.perf_counter()
    l, r = c_oracle.inner_join_i64(probe, build)
    dt = time.perf_counter() - t0
    extra = {}
    try:
        import pandas as pd
        m = min(n, 10_000_000)
        lp = pd.DataFrame({"k": probe[:m]})
        rp = pd.DataFrame({"k": build, "r": np.arange(len(build))})
        t0 = time.perf_counter()
        lp.merge(rp, on="k", how="inner")
        extra["pandas_merge_rows_per_s"] = m / (time.perf_counter() - t0)
        extra["pandas_rows"] = m
    except Exception as e:  # context only
        extra["pandas_error"] = repr(e)
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"probe {n} x build {len(build)} int64 rows, oracle/oracle.c orc_inner_join_i64 (count+retrieve)",
            "host_cpus": os.cpu_count(), "matches": int(len(l)), **extra}


def cpu_baseline_groupby(rows):
    import numpy as np
    from oracle import c_oracle
    n = int(rows)
    rng = np.random.default_rng(7)
    k = rng.integers(0, 1_000_000, n).astype(np.int32)
    v = rng.random(n)
    t0 = time.perf_counter()
    c_oracle.groupby_dense_sum_count(k, v, 1_000_000)
    dt = time.perf_counter() - t0
    extra = {}
    try:
        import pandas as pd
        m = min(n, 20_000_000)
        df = pd.DataFrame({"k": k[:m], "v": v[:m]})
        t0 = time.perf_counter()
        df.groupby("k", sort=False).agg(s=("v", "sum"), c=("v", "count"))
        extra["pandas_groupby_rows_per_s"] = m / (time.perf_counter() - t0)
        extra["pandas_rows"] = m
    except Exception as e:  # context only
        extra["pandas_error"] = repr(e)
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"{n} rows, 1e6 int32 groups, f64 sum+count, oracle/oracle.c orc_groupby_dense_sum_count",
            "host_cpus": os.cpu_count(), **extra}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from cudf_amd import Column, ops, _lib as L
    from cudf_amd.column import device_bytes, ptr, stream_ptr

    lib = L.lib
    n = int(args.rows)
    lib.gx_sort_set_algorithm(args.algo)
    lib.gx_groupby_set_algorithm(args.gb_algo, args.gb_split)
    lib.gx_sort_set_hybrid(0 if args.no_hybrid else 1)
    stream = stream_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    roofline = None
    extra = {}
    dist_step = None
    if world > 1:
        # distributed operators: same synthetic shards, one all-to-all exchange per step
        from cudf_amd import distributed as D
        local_ops = D.HipLocalOps()

        def as_tensor(col, dt):
            return col.data[: col.size * col.dtype.itemsize].view(dt)
        if args.workload in ("sort", "sorted_order"):
            dkeys = as_tensor(ops.random_column(np.int64, n, seed=42 + rank), torch.int64)
            dist_step = lambda: D.distributed_sort(dkeys, local=local_ops)
            dist_name = f"{n:.0e}-row-per-GPU int64 distributed sort (local sort, splitters, all-to-all, local sort)"
        elif args.workload == "join":
            nbr = max(1, n // 10)
            torch.manual_seed(12345 + rank)
            dbk = (torch.randperm(nbr, device="cuda") + rank * nbr) * 3 + 1           # globally distinct build keys
            dpk = as_tensor(ops.random_column(np.int64, n, seed=67890 + rank, lo=0, hi=int(nbr * world / 0.3)), torch.int64) * 3 + 1
            dist_step = lambda: D.distributed_inner_join(dpk, dbk, local=local_ops)
            dist_name = f"{n:.0e}-row-per-GPU probe x {nbr:.0e} build distributed inner join (hash partition, all-to-all, local join)"
        else:
            dgk = as_tensor(ops.random_column(np.int32, n, seed=7 + rank, lo=0, hi=1_000_000), torch.int32)
            dgv = as_tensor(ops.random_column(np.float64, n, seed=8 + rank), torch.float64)
            dist_step = lambda: D.distributed_groupby_sum_count(dgk, dgv, local=local_ops)
            dist_name = f"{n:.0e}-row-per-GPU groupby(int32 key, 1e6 groups).agg(f64 sum,count), partials exchanged"
    if args.workload in ("sort", "sorted_order"):
        if args.key_range:
            keys = ops.random_column(np.int64, n, seed=42 + rank, lo=args.key_range[0], hi=args.key_range[1])
        else:
            keys = ops.random_column(np.int64, n, seed=42 + rank)
        pairs = args.workload == "sorted_order"
        out = Column.empty(np.int32 if pairs else np.int64, n)
        nb = ctypes.c_size_t(0)
        if pairs:
            fn = lambda tmp, nbp: lib.gx_sorted_order(keys.gx, keys.data_ptr, None, n, 0, 0, 1, out.data_ptr, tmp, nbp, stream)
        else:
            fn = lambda tmp, nbp: lib.gx_sort_keys(keys.gx, keys.data_ptr, out.data_ptr, n, 0, tmp, nbp, stream)
        L.check(fn(None, ctypes.byref(nb)), "size query")
        tmp = device_bytes(nb.value)
        step = lambda: L.check(fn(ptr(tmp), ctypes.byref(nb)), "sort")
        bytes_per_row_pass = 24 if pairs else 16   # read key(+idx) + write key(+idx)
        model_bytes_row = 200 if pairs else 136    # 8-pass LSD model
        workload = f"{n:.0e}-row int64 " + ("sorted_order (radix sort pairs, int32 payload)" if pairs else "radix sort (cudf::sort, keys only)")
        if args.key_range:
            workload += f", keys uniform in [{args.key_range[0]}, {args.key_range[1]})"
        unit_rows = n
    elif args.workload == "join":
        nb_rows = max(1, n // 10)
        # build: distinct keys (a permutation-like bijection of iota), probe: 30% hit rate
        # (cpp/benchmarks/join/generate_input_tables.cu:24-103: unique build keys, selectivity 0.3)
        bk = Column.empty(np.int64, nb_rows)
        bkt = bk.data[: nb_rows * 8].view(torch.int64)
        torch.manual_seed(12345 + rank)
        bkt.copy_(torch.randperm(nb_rows, device="cuda") * 3 + 1)  # distinct keys {3i+1}, shuffled
        pk = ops.random_column(np.int64, n, seed=67890 + rank, lo=0, hi=int(nb_rows / 0.3))
        pk.data[: n * 8].view(torch.int64).mul_(3).add_(1)          # hits a build key w.p. 0.3
        hj = ops.HashJoin(bk)  # warm-up build (allocations, module load)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        hj = ops.HashJoin(bk)
        torch.cuda.synchronize()
        extra["join_build_ms"] = (time.perf_counter() - tb) * 1e3  # build of the 1e8-row side (not in `value`)
        lo = Column.empty(np.int32, n)
        ro = Column.empty(np.int32, n)
        cur = torch.zeros(1, dtype=torch.int64, device="cuda")

        part_bits = lib.gx_join_partition_bits(8, hj.table_bytes) if not args.no_partitioned_join else 0
        extra["join_partition_bits"] = part_bits
        jnb = ctypes.c_size_t(0)
        if part_bits:
            L.check(lib.gx_join_probe_partitioned(8, pk.data_ptr, n, ptr(hj.table), hj.table_bytes, 0, lo.data_ptr,
                                                  ro.data_ptr, n, ptr(cur), None, ctypes.byref(jnb), stream), "query")
            jtmp = device_bytes(jnb.value)

        def step():
            cur.zero_()
            if part_bits:
                L.check(lib.gx_join_probe_partitioned(8, pk.data_ptr, n, ptr(hj.table), hj.table_bytes, 0, lo.data_ptr,
                                                      ro.data_ptr, n, ptr(cur), ptr(jtmp), ctypes.byref(jnb), stream), "probe")
            else:
                L.check(lib.gx_join_probe(8, pk.data_ptr, None, n, ptr(hj.table), hj.table_bytes, 0, lo.data_ptr,
                                          ro.data_ptr, n, ptr(cur), stream), "probe")
        workload = f"{n:.0e}-row int64 probe x {nb_rows:.0e}-row build inner hash join (probe phase timed)"
        unit_rows = n
    elif args.workload in ("reduce", "scan", "gather"):
        # single streaming kernels of the path: f64 SUM reduce, int64 inclusive
        # SUM scan, 8-byte gather through a random int32 map
        src = ops.random_column(np.float64 if args.workload == "reduce" else np.int64, n, seed=11 + rank)
        nb = ctypes.c_size_t(0)
        if args.workload == "reduce":
            res = Column.empty(np.float64, 1)
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            fn = lambda tmp, nbp: lib.gx_reduce(src.gx, src.data_ptr, None, n, L.OP_SUM, L.FLOAT64, res.data_ptr, ptr(cnt), tmp, nbp, stream)
            bpr_simple, kname = 8, "gx_reduce f64 SUM (k_chunk_reduce)"
        elif args.workload == "scan":
            dst = Column.empty(np.int64, n)
            fn = lambda tmp, nbp: lib.gx_scan(src.gx, src.data_ptr, None, n, L.OP_SUM, 1, dst.data_ptr, tmp, nbp, stream)
            bpr_simple, kname = 16, "gx_scan int64 inclusive SUM (reduce-then-scan, 3 launches; 24 B/row moved)"
        else:
            gm = ops.random_column(np.int32, n, seed=13 + rank, lo=0, hi=n)
            dst = Column.empty(np.int64, n)
            fn = None
            bpr_simple, kname = 20, "gx_gather 8-byte rows through a uniform random int32 map"
        if fn is not None:
            L.check(fn(None, ctypes.byref(nb)), "size query")
            tmp = device_bytes(nb.value)
            step = lambda: L.check(fn(ptr(tmp), ctypes.byref(nb)), args.workload)
        else:
            step = lambda: L.check(lib.gx_gather(8, src.data_ptr, None, n, gm.data_ptr, n, 0, dst.data_ptr, None, stream), "gather")
        workload = f"{n:.0e}-row {kname}"
        unit_rows = n
    else:  # groupby
        gk = ops.random_column(np.int32, n, seed=7 + rank, lo=0, hi=1_000_000)
        gv = ops.random_column(np.float64, n, seed=8 + rank)
        mg = 1 << 20
        ok, osum = Column.empty(np.int32, mg), Column.empty(np.float64, mg)
        ocv = Column.empty(np.int32, mg)
        ng = torch.zeros(1, dtype=torch.int64, device="cuda")
        nb = ctypes.c_size_t(0)
        fn = lambda tmp, nbp: lib.gx_groupby_sum_count(gk.gx, gk.data_ptr, None, gv.gx, gv.data_ptr, None, n, mg,
                                                       ok.data_ptr, osum.data_ptr, ocv.data_ptr, None, ptr(ng), tmp, nbp, stream)
        L.check(fn(None, ctypes.byref(nb)), "size query")
        tmp = device_bytes(nb.value)
        step = lambda: L.check(fn(ptr(tmp), ctypes.byref(nb)), "groupby")
        workload = f"{n:.0e}-row groupby(int32 key, 1e6 groups).agg(float64 sum,count)"
        unit_rows = n

    single_step = step
    if dist_step is not None:
        step = dist_step
        workload = dist_name
    for _ in range(args.warmup):
        step()
    if args.workload in ("sort", "sorted_order"):
        lib.gx_sort_profile(1)
    barrier()
    t0 = time.perf_counter()
    pass_ms_acc, hist_ms_acc, launches = 0.0, 0.0, 0
    hyb_acc, hyb_n = [0.0, 0.0, 0.0, 0.0], 0
    for _ in range(args.steps):
        step()
        if args.workload in ("sort", "sorted_order") and dist_step is None:
            # reading the events waits for this step only; it is part of the timed region
            h = ctypes.c_float()
            p = (ctypes.c_float * 8)()
            k = ctypes.c_int()
            L.check(lib.gx_sort_profile_read(ctypes.byref(h), p, ctypes.byref(k)), "profile_read")
            act = [x for x in list(p)[: k.value] if x > 0.05]  # skipped passes exit in microseconds
            pass_ms_acc += sum(act)
            launches += len(act)
            hist_ms_acc += h.value
            h4 = (ctypes.c_float * 4)()
            if lib.gx_sort_profile_read_hybrid(h4) == 0:
                for i in range(4):
                    hyb_acc[i] += h4[i]
                hyb_n += 1
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # correctness guard on the timed output (device-side, cheap): sorted + same multiset
    if dist_step is not None:
        if args.workload in ("sort", "sorted_order"):
            res = dist_step()
            assert bool((res[1:] >= res[:-1]).all()), "distributed sort: shard not sorted"
            tot = torch.tensor([res.numel()], device="cuda", dtype=torch.int64)
            dist.all_reduce(tot)
            assert int(tot.item()) == n * world, "distributed sort lost rows"
            del res
        if args.workload == "sort":
            # roofline of the dominant LOCAL kernel: one profiled single-GPU sort of this rank's shard,
            # outside the timed region
            single_step()
            h = ctypes.c_float()
            p8 = (ctypes.c_float * 8)()
            k = ctypes.c_int()
            L.check(lib.gx_sort_profile_read(ctypes.byref(h), p8, ctypes.byref(k)), "profile_read")
            act = [x for x in list(p8)[: k.value] if x > 0.05]
            pass_ms_acc, launches, hist_ms_acc = sum(act), len(act), h.value * args.steps
            h4 = (ctypes.c_float * 4)()
            if lib.gx_sort_profile_read_hybrid(h4) == 0:
                hyb_acc, hyb_n = list(h4), 1
    elif args.workload == "sort":
        cin, cout = ops.checksum(keys), ops.checksum(out)
        assert cout[2] == 0 and cin[:2] == cout[:2], "sort output invalid"
        st = ctypes.c_int(0)
        lib.gx_sort_status(ptr(tmp), ctypes.byref(st), stream)
        assert st.value == 0, "look-back timed out"
    sort_info = None
    local_sort_ms = ms_per_step  # duration the whole-sort model figures refer to
    if dist_step is not None and args.workload == "sort":
        local_sort_ms = hist_ms_acc / args.steps + (sum(hyb_acc) if hyb_n else pass_ms_acc)
    if args.workload in ("sort", "sorted_order") and (dist_step is None or args.workload == "sort"):
        info = (ctypes.c_int32 * 8)()
        lib.gx_sort_info(ptr(tmp), info, stream)
        sort_info = dict(zip(["hybrid_attempted", "hybrid_used", "d1", "shift2", "bits2", "lds_passes", "max_cell",
                              "lsd_passes"], list(info)))
    if args.workload in ("sort", "sorted_order") and sort_info and sort_info["hybrid_used"] and hyb_n:
        # hybrid MSD path: per-kernel algorithmic bytes: partition passes and the local
        # sort read 8 + write 8 B/row, the joint histogram reads 8 B/row
        ms = [x / hyb_n for x in hyb_acc]
        names = ["k_msd_pass level 0 (8-bit partition, 8 XCD chains)", "k_hist2+k_plan2 (joint histogram)",
                 "k_msd_pass level 1 (partition inside buckets)", "k_local_sort (LDS sort of <=16384-key cells)"]
        bpr = [20, 8, 24, 24] if args.workload == "sorted_order" else [16, 8, 16, 16]  # pairs carry a 4-B index
        dom = max(range(4), key=lambda i: ms[i])
        achieved = bpr[dom] * n / (ms[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": bpr[dom] * n,
                    "avg_launch_ms": ms[dom], "launches_per_step": 1.0,
                    "kernels_ms": dict(zip(names, ms)), "kernels_GBps": {k: b * n / (m * 1e-3) / 1e9 for k, b, m in zip(names, bpr, ms)},
                    "hist_kernel_ms": hist_ms_acc / args.steps,
                    "path_bytes_per_row": 8 + sum(bpr), "path_GBps": (8 + sum(bpr)) * n / (local_sort_ms * 1e-3) / 1e9,
                    "whole_sort_model_GBps": model_bytes_row * n / (local_sort_ms * 1e-3) / 1e9,
                    "whole_sort_model_frac": model_bytes_row * n / (local_sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "sort_info": sort_info}
    elif args.workload in ("sort", "sorted_order") and launches:
        avg_ms = pass_ms_acc / launches
        achieved = bytes_per_row_pass * n / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_radix_pass (one 8-bit digit scatter pass)", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": bytes_per_row_pass * n, "avg_launch_ms": avg_ms,
                    "launches_per_step": launches / args.steps,
                    "hist_kernel_ms": hist_ms_acc / args.steps,
                    "whole_sort_model_GBps": model_bytes_row * n / (local_sort_ms * 1e-3) / 1e9,
                    "whole_sort_model_frac": model_bytes_row * n / (local_sort_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "sort_info": sort_info}
    elif dist_step is not None:
        roofline = None  # distributed join / groupby: see the N=1 lines for the kernels' rooflines
    elif args.workload == "join":
        matches = int(cur.item())
        algb = 24 * n + 16 * matches
        ach = algb / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_pj_hist+k_pj_scatter+k_pj_probe (partitioned probe)" if extra.get("join_partition_bits") else "k_probe", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": algb,
                    "avg_launch_ms": ms_per_step, "matches": matches}
    elif args.workload in ("reduce", "scan", "gather"):
        ach = bpr_simple * n / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": bpr_simple * n,
                    "avg_launch_ms": ms_per_step}
    elif args.workload == "groupby":
        ach = 12 * n / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_aggregate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": 12 * n,
                    "avg_launch_ms": ms_per_step, "groups": int(ng.item())}

    # HBM bytes of the dominant kernel from the committed PMC passes (same command, 1e9 rows)
    if roofline is not None and n == 1_000_000_000:
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic_1e9.json")))
            for key, v in tr["kernels"].items():
                if roofline["kernel"].startswith(key):
                    roofline["traffic"] = v["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = tr["source"] + "; " + tr["correction"]
        except (OSError, KeyError, ValueError):
            pass
    if rank == 0:
        cpu = None
        if args.cpu and world == 1:
            # bounded sample: ~10-30 s of single-thread CPU work
            cpu_rows = args.cpu_rows or {"sort": 5e8, "sorted_order": 5e8, "join": 2e8, "groupby": 5e8}.get(args.workload, 1e8)
            fnc = {"sort": cpu_baseline_sort, "sorted_order": cpu_baseline_sort, "join": cpu_baseline_join,
                   "groupby": cpu_baseline_groupby}.get(args.workload)
            cpu = fnc(cpu_rows) if fnc else None
        line = {
            "metric": "rows/sec + achieved HBM GB/s: 1e9-row int64 sort & hash-join, 1/2/4/8 GPU",
            "value": unit_rows * world / (dt / args.steps), "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"sort": "int64", "sorted_order": "int64", "join": "int64", "groupby": "f64", "reduce": "f64",
                                          "scan": "int64", "gather": "int64"}[args.workload],
            "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": n, "algo": args.algo, "gb_algo": args.gb_algo,
                       "parallelism": (f"{world} ranks, row shards, one all-to-all exchange per step (RCCL over xGMI)"
                                       if world > 1 else "1 GPU")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
