"""Analytic model of the splitter mode's cells (gx_sort.hip k_sp_plan / k_hf_plan stage 2): which share of a 1e9-key column lands in
cells that overflow their slot (-> the big-cell path X), under (a) equal-width cells sized by the median-derived peak factor (round 5
as measured in run 9 / 11) and (b) cells cut on a piecewise-linear warp of the bucket's fraction, measured by the n/32 sample.
Expected counts come from the true CDF of each distribution; a cell counts as overflowing when expected + 3 sigma > its slot."""
import sys
import numpy as np
from scipy import stats

N = 1_000_000_000
NS = 16384
CELL = 7400.0
CMAX = 8192
NCMAX = 1024
PIECES = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def qpos():
    step, nreg = 68, (NS - 64) // 68
    t = np.arange(nreg + 10)
    return np.where(t < 5, 2 << np.minimum(t, 30), np.where(t < 5 + nreg, step * (t - 4), NS - (32 >> np.clip(t - 5 - nreg, 0, 5))))


def cap_of(m):
    c = m + 6.0 * np.sqrt(m) + 64.0
    return np.minimum(np.ceil(c / 16) * 16, CMAX)


def run(name, dist, rng):
    """dist: frozen scipy distribution over the KEY axis (monotone image of the sortable key)"""
    samp = np.sort(dist.rvs(NS, random_state=rng))
    T = np.unique(samp[qpos()])
    lo = np.concatenate([[samp[0]], T])
    hi = np.concatenate([T, [samp[-1]]])
    nb = len(lo)
    # exact bucket masses (the first / last bucket hold the tails beyond the sample's range as well)
    tl = np.concatenate([[-np.inf], T]); th = np.concatenate([T, [np.inf]])
    cc = N * (dist.cdf(th) - dist.cdf(tl))
    w = hi - lo
    out = {}
    # ---- (a) equal-width cells, peak factor from the sampled median
    ia = np.searchsorted(samp, lo, side="left"); ib = np.searchsorted(samp, hi, side="left"); ib[-1] = NS
    pf = np.ones(nb)
    for b in range(nb):
        if ib[b] - ia[b] >= 8:
            f = (samp[(ia[b] + ib[b]) >> 1] - lo[b]) / w[b]
            f = min(f, 1 - f); f = max(f, 0.2)
            pf[b] = min(max((0.5 - f * f) / (f * (1 - f)) + 0.1, 1.0), 2.5)
    def overflow(edges_fn, nc, capmean):
        over = 0.0; cells = 0; big = 0
        for b in range(nb):
            e = edges_fn(b, int(nc[b]))
            e[0] = tl[b]; e[-1] = th[b]
            cnt = N * np.diff(dist.cdf(e))
            cap = capmean[b]
            bad = cnt + 3.0 * np.sqrt(cnt) > cap
            over += cnt[bad].sum(); big += bad.sum(); cells += int(nc[b])
        return over / N, cells, big
    nc = np.clip(np.ceil(cc * pf / CELL), 1, NCMAX)
    mean = cc / nc + 1
    m3 = np.maximum(mean, np.maximum(np.roll(mean, 1), np.roll(mean, -1)))
    out["a_equal_width"] = overflow(lambda b, k: lo[b] + w[b] * np.arange(k + 1) / k, nc, cap_of(m3))
    out["a_cap_at_peak"] = overflow(lambda b, k: lo[b] + w[b] * np.arange(k + 1) / k, nc, cap_of(cc * pf / nc))
    # ---- (b) warp: PIECES equal-width pieces per bucket, masses from the n / 32 sample (binomial noise), floor 1/64 of the even share
    d = np.zeros((nb, PIECES))
    peak = np.ones(nb)
    for b in range(nb):
        e = lo[b] + w[b] * np.arange(PIECES + 1) / PIECES
        e[0] = tl[b]; e[-1] = th[b]
        m = np.diff(dist.cdf(e)) * N / 32
        m = rng.poisson(m).astype(float)
        m = np.maximum(m, m.sum() / PIECES / 64)
        d[b] = m / m.sum()
        # residual skew inside a piece: density taken as linear between the neighbouring pieces' means
        ext = np.concatenate([[d[b][0]], d[b], [d[b][-1]]])
        r = np.maximum((ext[:-2] + ext[1:-1]) / 2, (ext[2:] + ext[1:-1]) / 2) / ext[1:-1]
        peak[b] = min(np.max(r[d[b] > 0.25 / PIECES]) if np.any(d[b] > 0.25 / PIECES) else 1.0, 2.5)
    def warp_edges(b, k):
        Y = np.concatenate([[0], np.cumsum(d[b])]); Y[-1] = 1.0
        y = np.arange(k + 1) / k
        j = np.clip(np.searchsorted(Y, y, side="right") - 1, 0, PIECES - 1)
        frac = (j + (y - Y[j]) / d[b][j]) / PIECES
        return lo[b] + w[b] * np.clip(frac, 0, 1)
    for label, margin in (("b_warp_m1.00", 1.0), ("b_warp_m1.03", 1.03), ("b_warp_peak", None)):
        p = peak if margin is None else np.full(nb, margin)
        nc = np.clip(np.ceil(cc * p / CELL), 1, NCMAX)
        out[label] = overflow(warp_edges, nc, cap_of(cc * p / nc))
    print(f"{name:14s} buckets {nb:3d}  median pf {np.median(pf):.2f}  median residual peak {np.median(peak):.3f}")
    for k, (o, cells, big) in out.items():
        print(f"    {k:16s} keys in overflowing cells {100 * o:6.2f} %   cells {cells:7d}  big {big:5d}")


rng = np.random.default_rng(5)
run("normal", stats.norm(0, 1), rng)
run("lognormal", stats.lognorm(1.0), rng)
run("exp (zipf-ish)", stats.expon(), rng)
run("uniform", stats.uniform(0, 1), rng)
# float64 keys: the sortable form is monotone in x but NOT linear: model the key axis as sign * (exponent + mantissa fraction), i.e.
# k(x) = sign(x) * (log2|x| piecewise-linear); use a distribution over k by transforming samples numerically
class Mapped:
    """distribution of k = g(x) for x ~ base, g monotone increasing (given with its inverse)"""
    def __init__(self, base, g, ginv): self.base, self.g, self.ginv = base, g, ginv
    def rvs(self, n, random_state=None): return self.g(self.base.rvs(n, random_state=random_state))
    def cdf(self, k): return self.base.cdf(self.ginv(np.asarray(k, dtype=float)))
def fkey(x):  # IEEE double bits as a real number: (exponent + mantissa fraction), mirrored for negatives; offset so that 2^-60 ~ 0
    x = np.asarray(x, dtype=float); a = np.maximum(np.abs(x), 2.0 ** -60)
    e = np.floor(np.log2(a)); k = e + (a / 2.0 ** e - 1.0) + 61.0
    return np.sign(x) * k
def fkey_inv(k):
    k = np.asarray(k, dtype=float); a = np.abs(k) - 61.0
    with np.errstate(over="ignore", invalid="ignore"):
        e = np.floor(a); x = 2.0 ** e * (1.0 + (a - e))
        x = np.where(np.abs(k) < 1.0, 0.0, x)
    return np.where(np.isinf(k), k, np.sign(k) * x)
run("f64 N(0,1)", Mapped(stats.norm(0, 1), fkey, fkey_inv), rng)
run("f64 U[0,1)", Mapped(stats.uniform(0, 1), fkey, fkey_inv), rng)
