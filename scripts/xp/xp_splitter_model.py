#!/usr/bin/env python3
"""CPU model of the sort DESIGN.md section 8 row 1 proposes (NOT built): sample-chosen SPLITTERS at level 0, per-bucket shifted
bit digits at level 1, 8192-key cells, cells that outgrow their slot through X (the big-cell path that exists).

What it answers, per key distribution, before any kernel is written:
  * how uneven are the 256 level-0 buckets when the 255 splitters are quantiles of an m-key sample (m = 16 K fits one workgroup's
    LDS sort; 64 K needs a small multi-pass sort) -- this fixes how many level-1 bits the fullest bucket needs;
  * with level 1 of bucket b on ((key - splitter_b) >> sh_b), sh_b from the bucket's WIDTH: what fraction of the keys lands in
    cells above 8192 keys (they go through X: must stay well below half the column) and how many cells are used;
  * the same for today's bit digits (top 8 varying bits, then bits2 bits), i.e. the path that is declined today.
Keys are modelled as unsigned 64-bit SORTABLE words (signed ints: sign flipped; doubles: cub's order-preserving map).  The model
runs at n = 2^24 keys with 8192-key cells (bits2 = 3) -- the cell count per bucket is smaller than at 1e9 rows (2^9), the
per-bucket density structure is the same; numbers that depend on it are flagged in the output.

usage: python scripts/xp/xp_splitter_model.py [--n 16777216] [--sample 16384 65536]
"""
import argparse

import numpy as np

CELL = 8192


def sortable_i64(v):
    return v.astype(np.int64).view(np.uint64) ^ np.uint64(1 << 63)


def sortable_f64(x):
    b = x.astype(np.float64).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    return np.where(neg, ~b, b | np.uint64(1 << 63))


def distributions(n, rng):
    yield "uniform 64-bit", sortable_i64(rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64))
    yield "uniform [0, 1e12)", sortable_i64(rng.integers(0, 10**12, n, dtype=np.int64))
    yield "normal(0, 2^40) ints", sortable_i64(np.round(rng.standard_normal(n) * 2.0**40))
    yield "lognormal(mu 10, sigma 2) ints", sortable_i64(np.minimum(np.round(np.exp(10 + 2 * rng.standard_normal(n))), 2.0**62))
    yield "exponential(1e9) ints", sortable_i64(np.round(rng.exponential(1e9, n)))
    u = np.maximum(rng.random(n), 2.0**-53)
    yield "Zipf-like floor(u^-5) <= 2^31", sortable_i64(np.minimum(np.floor(u**-5.0), 2.0**31))
    yield "float64 N(0, 1)", sortable_f64(rng.standard_normal(n))
    yield "float64 U[0, 1)", sortable_f64(rng.random(n))
    ts = 1_700_000_000_000 + np.cumsum(rng.exponential(40.0, n)).astype(np.int64)          # event times, ms
    yield "timestamps (Poisson arrivals)", sortable_i64(rng.permutation(ts))
    c = rng.integers(0, 1 << 20, n, dtype=np.int64) + np.where(rng.random(n) < 0.5, np.int64(1) << 60, np.int64(0))
    yield "two narrow clusters 2^60 apart", sortable_i64(c)
    hot = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    hot[rng.random(n) < 0.10] = 1234567890123
    yield "uniform + 10 % one value", sortable_i64(hot)


def cells_report(bucket, rel, width_bits, counts, bits2_cap, widths=None, fill=0.90):
    """level 1 inside every bucket.  widths is None: today's BIT digit -- bits2_b bits below the bucket's top bit, as many as the
    bucket's mean cell needs (capped).  widths given (the splitter sort): an EQUAL-WIDTH split into ncell_b = count / (fill * CELL)
    cells (any integer up to 2^cap: cell = rel * ncell_b / width_b, a multiply-high), so a bucket whose width is not a power of
    two still fills all of its cells.  Cells whose keys are all equal need no sort (heavy hitters: equality cells) and do not
    count as big.  Returns (share of keys in cells above CELL that are not constant, cells used, most cells in a bucket)."""
    big_keys, cells_used, max_cells = 0, 0, 0
    order = np.argsort(bucket, kind="stable")
    starts = np.concatenate([[0], np.cumsum(counts)])
    for b in range(len(counts)):
        c = int(counts[b])
        if c == 0:
            continue
        r = rel[order[starts[b]:starts[b + 1]]]
        if widths is None:
            need = 0
            while (c >> need) > 0.955 * CELL and need < bits2_cap:
                need += 1
            wb = int(width_bits[b])
            b2 = min(need, wb)
            ncell = 1 << b2
            cell = (r >> np.uint64(wb - b2)).astype(np.int64) if wb > b2 else r.astype(np.int64)
        else:
            w = float(widths[b])
            ncell = int(min(max(1, -(-c // int(fill * CELL))), 1 << bits2_cap, max(1.0, w)))
            cell = np.minimum((r.astype(np.float64) * (ncell / w)).astype(np.int64), ncell - 1)   # (the kernel: 64-bit multiply-high)
        max_cells = max(max_cells, ncell)
        cc = np.bincount(cell, minlength=ncell)
        over = np.nonzero(cc > CELL)[0]
        if len(over):
            # a cell of one repeated value is an equality cell
            so = np.argsort(cell, kind="stable")
            cs = np.concatenate([[0], np.cumsum(cc)])
            for k in over:
                rr = r[so[cs[k]:cs[k + 1]]]
                if rr.min() != rr.max():
                    big_keys += int(cc[k])
        cells_used += int((cc > 0).sum())
    return big_keys, cells_used, max_cells


def model_splitters(s, m, rng, bits2_cap=10):
    n = len(s)
    samp = np.sort(s[rng.integers(0, n, m)])
    q = samp[(np.arange(1, 256) * m) // 256]                       # 255 quantiles of the sample
    vals, reps = np.unique(q, return_counts=True)
    heavy = vals[reps >= 2]                                         # a value that fills two quantiles holds > 1/256 of the column:
    sp = np.unique(np.concatenate([vals, heavy + np.uint64(1)]))    # ... it gets a bucket of its own, [v, v + 1)
    bucket = np.searchsorted(sp, s, side="right")
    nb = len(sp) + 1
    counts = np.bincount(bucket, minlength=nb)
    lo = np.concatenate([[s.min()], sp])
    hi = np.concatenate([sp, [s.max() + np.uint64(1)]])
    width = (hi - lo).astype(np.float64)
    rel = s - lo[bucket]
    big, used, mc = cells_report(bucket, rel, None, counts, bits2_cap, widths=np.maximum(width, 1.0))
    ordinary = counts[width > 1.5]
    return nb, (ordinary.max() if len(ordinary) else 0) / (n / 256.0), big / n, used, mc, float(counts[width <= 1.5].sum()) / n


def model_bit_digits(s, bits2_cap=10):
    n = len(s)
    V = int(np.bitwise_or.reduce(s)) & int(np.bitwise_or.reduce(~s))
    if V == 0:
        return 1, 256.0, 0.0, 1, 0
    top = V.bit_length() - 1
    shift0 = max(top - 7, 0)
    bucket = ((s >> np.uint64(shift0)) & np.uint64(0xFF)).astype(np.int64)
    counts = np.bincount(bucket, minlength=256)
    rel = s & np.uint64((1 << shift0) - 1)
    width_bits = np.full(256, shift0, np.int64)
    big, used, mc = cells_report(bucket, rel, width_bits, counts, bits2_cap)
    return int((counts > 0).sum()), counts.max() / (n / 256.0), big / n, used, mc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 24)
    ap.add_argument("--sample", type=int, nargs="+", default=[16384, 65536])
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    print(f"# n = {a.n} keys, {CELL}-key cells, at most 2^10 cells per bucket; 'big' = keys in cells above {CELL} whose keys are not all equal (sorted through X)")
    print(f"# {'distribution':34s} | {'BIT DIGITS (today): buckets, fullest / mean, big':50s} | " + " | ".join(f"SPLITTERS from a {m}-key sample: buckets, fullest ordinary / mean, in equality buckets, big, most cells" for m in a.sample))
    for name, s in distributions(a.n, rng):
        nb, mx, big, used, mc = model_bit_digits(s)
        row = f"{name:36s} | {nb:4d} buckets, {mx:7.2f}x, big {100 * big:5.1f} %               "
        for m in a.sample:
            nb2, mx2, big2, used2, mc2, eq = model_splitters(s, m, rng)
            row += f" | {nb2:3d} buckets, {mx2:5.2f}x, eq {100 * eq:5.1f} %, big {100 * big2:5.1f} %, cells <= {mc2:4d}"
        print(row)


if __name__ == "__main__":
    main()
