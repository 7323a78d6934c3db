"""CPU checks of the bit-level formulas the HIP kernels rely on, restated in Python and compared with
brute force.  They do not run the kernels (the -m gpu tests do); they pin the arithmetic the kernels
were derived from, so a future edit of a constant or a shift can be checked without a GPU.

  * scan_tags8 (cudf_amd/csrc/gx_join.hip): the 4-bit slot tags of the partitioned hash-join probe --
    carry-free zero-nibble detection, "tag matches before the first empty slot";
  * k_bitmask_copy (cudf_amd/csrc/gx_gather.hip): funnel shift + edge-word masks of the offset bitmap copy;
  * normalise_float / gx_pack_keys (cudf_amd/csrc/gx_rank.hip): float equality classes and key packing;
  * mm_encode order (cudf_amd/csrc/gx_groupby.hip) == oracle.sortable_bits order for floats.
"""
import numpy as np

from oracle import cudf_oracle as orc

M32 = 0xFFFFFFFF


def _scan_tags8(w0, w1, li, tagpat):
    """Python transcription of scan_tags8: returns (cand mask, chain_ends_in_window)."""
    two = (w1 << 32) | w0
    x = (two >> ((li & 7) * 4)) & M32                               # v_alignbit_b32(w1, w0, shift)
    y = x ^ tagpat
    z = ~(((x & 0x77777777) + 0x77777777) | x) & 0x88888888 & M32
    m = ~(((y & 0x77777777) + 0x77777777) | y) & 0x88888888 & M32
    cand = m & (((z & (-z & M32)) - 1) & M32)
    return cand, z != 0


def test_scan_tags8_matches_bruteforce():
    rng = np.random.default_rng(0)
    for _ in range(20000):
        tags = rng.integers(0, 16, 16)
        if rng.random() < 0.5:
            tags[rng.integers(0, 16, rng.integers(0, 6))] = 0      # some empty slots
        w0 = sum(int(t) << (4 * i) for i, t in enumerate(tags[:8]))
        w1 = sum(int(t) << (4 * i) for i, t in enumerate(tags[8:]))
        li = int(rng.integers(0, 8))
        tag = int(rng.integers(1, 16))
        cand, ended = _scan_tags8(w0, w1, li, tag * 0x11111111)
        window = tags[li:li + 8]
        exp_cand, exp_ended = 0, False
        for i, t in enumerate(window):
            if t == 0:
                exp_ended = True
                break
            if t == tag:
                exp_cand |= 0x8 << (4 * i)                          # flag = top bit of the nibble
        assert cand == exp_cand and ended == exp_ended


def test_bitmask_copy_word_formula():
    """dst word = funnel(src[sw], src[sw + 1]) >> sh, masked at the two edge words."""
    rng = np.random.default_rng(1)
    for _ in range(300):
        nbits = int(rng.integers(1, 200))
        doff, soff = int(rng.integers(0, 70)), int(rng.integers(0, 70))
        src = rng.random(soff + nbits) > 0.5
        dst = rng.random(doff + nbits + 40) > 0.5
        def words(bits):  # LSB-first uint32 words, padded with two spare words
            padded = np.zeros(((len(bits) + 31) // 32 + 2) * 32, bool)
            padded[: len(bits)] = bits
            return np.packbits(padded, bitorder="little").view(np.uint32).astype(np.uint64)
        sw, dw = words(src), words(dst)
        send = (soff + nbits + 31) >> 5
        for w in range(doff >> 5, (doff + nbits + 31) >> 5):
            lo, hi = max(w << 5, doff), min((w + 1) << 5, doff + nbits)
            mask = 0xFFFFFFFF if hi - lo == 32 else (((1 << (hi - lo)) - 1) << (lo & 31))
            sb = soff + (lo - doff)
            two = int(sw[sb >> 5])
            if (sb & 31) and (sb >> 5) + 1 < send:
                two |= int(sw[(sb >> 5) + 1]) << 32
            v = ((two >> (sb & 31)) << (lo & 31)) & M32
            dw[w] = (int(dw[w]) & ~mask & M32) | (v & mask)
        got = np.unpackbits(dw.astype(np.uint32).view(np.uint8), bitorder="little")[: len(dst)].astype(bool)
        exp = dst.copy()
        exp[doff:doff + nbits] = src[soff:soff + nbits]
        np.testing.assert_array_equal(got, exp)


def _normalise(bits, size):
    if size == 4:
        x = bits & 0x7FFFFFFF
        return 0 if x == 0 else (0x7FC00000 if x > 0x7F800000 else bits)
    x = bits & 0x7FFFFFFFFFFFFFFF
    return 0 if x == 0 else (0x7FF8000000000000 if x > 0x7FF0000000000000 else bits)


def test_float_normalisation_gives_the_row_comparators_equality_classes():
    """normalise_float(a) == normalise_float(b)  <=>  a == b or both NaN  (so -0.0 == +0.0, NaN == NaN)."""
    for dt, size in ((np.float32, 4), (np.float64, 8)):
        u = np.uint32 if size == 4 else np.uint64
        vals = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, -np.nan, 1e-40, -1e-40, 3.5], dt)
        bits = vals.view(u)
        extra = np.array([0x7FC00001, 0xFFC12345], np.uint32) if size == 4 else np.array([0x7FF8000000000001, 0xFFF0000000000001], np.uint64)
        vals = np.concatenate([vals, extra.view(dt)])
        bits = np.concatenate([bits, extra])
        for a, ba in zip(vals, bits):
            for b, bb in zip(vals, bits):
                same = (a == b) or (np.isnan(a) and np.isnan(b))
                assert (_normalise(int(ba), size) == _normalise(int(bb), size)) == bool(same)
        # and the oracle's join key uses the same classes
        k = orc._join_key(vals)
        for i in range(len(vals)):
            for j in range(len(vals)):
                assert (k[i] == k[j]) == (_normalise(int(bits[i]), size) == _normalise(int(bits[j]), size))


def test_pack_keys_layout():
    """column 0 most significant; widths sum to <= 8 bytes; equality of packed words == equality of rows."""
    rng = np.random.default_rng(2)
    a = rng.integers(-5, 5, 200).astype(np.int32)
    b = rng.integers(0, 4, 200).astype(np.uint16)
    c = rng.integers(-3, 3, 200).astype(np.int8)
    packed = (a.view(np.uint32).astype(np.uint64) << np.uint64(24)) | (b.astype(np.uint64) << np.uint64(8)) | c.view(np.uint8).astype(np.uint64)
    rows = list(zip(a.tolist(), b.tolist(), c.tolist()))
    for i in range(0, 200, 7):
        for j in range(200):
            assert (packed[i] == packed[j]) == (rows[i] == rows[j])


def test_minmax_word_order_is_the_sort_order():
    """gx_groupby_min_max widens values to 64-bit words whose unsigned order is the value order (sign flip for
    integers, IEEE total-order flip with -0.0 -> +0.0 and NaN above +Inf for floats): the same map as
    oracle.sortable_bits, which the sort goldens pin."""
    f = np.array([-np.inf, -3.5, -0.0, 0.0, 1e-300, 2.0, np.inf, np.nan], np.float64)
    sb = orc.sortable_bits(f)
    assert sb[2] == sb[3]                                            # -0.0 == +0.0
    assert all(sb[i] <= sb[i + 1] for i in range(len(sb) - 1)) and sb[-1] == np.uint64(2**64 - 1)
    i = np.array([-2**63, -1, 0, 1, 2**63 - 1], np.int64)
    si = orc.sortable_bits(i)
    assert all(si[k] < si[k + 1] for k in range(len(si) - 1))


def test_split_sort_four_transposition_phases_suffice():
    """wave_split_sort (gx_common.hpp): after the counting split on one byte every bin holds <= 4 keys and
    the bins are in order; the kernel then runs even, odd, even, odd compare-exchange phases over the
    whole wave (element pairs (2l, 2l+1), then (2l+1, 2l+2)).  Exhaustively: every arrangement of bins of
    1..4 keys, at either alignment, in every internal order, comes out sorted."""
    import itertools

    def phases(a):
        a = list(a)
        for _ in range(2):
            for i in range(0, len(a) - 1, 2):      # even phase
                if a[i + 1] < a[i]:
                    a[i], a[i + 1] = a[i + 1], a[i]
            for i in range(1, len(a) - 1, 2):      # odd phase
                if a[i + 1] < a[i]:
                    a[i], a[i + 1] = a[i + 1], a[i]
        return a

    checked = 0
    for lead in (0, 1):                            # a 1-key bin in front shifts the alignment
        for sizes in itertools.product((1, 2, 3, 4), repeat=3):
            bins = ([[0]] if lead else []) + [list(range(10 * (b + 1), 10 * (b + 1) + s)) for b, s in enumerate(sizes)]
            for perm in itertools.product(*[itertools.permutations(b) for b in bins]):
                arr = [x for b in perm for x in b]
                assert phases(arr) == sorted(arr)
                checked += 1
    assert checked > 10000
    # and five keys in one bin are NOT always sorted by four phases: the kernel's "any bin > 4 -> network" guard
    assert any(phases(p) != sorted(p) for p in itertools.permutations(range(5)))


def _network_steps(name):
    """The (xor distance, low-bit) compare-exchange steps of a wave network, parsed from gx_common.hpp."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cudf_amd", "csrc", "gx_common.hpp")).read()
    body = re.search(r"void " + name + r"\(uint64_t& k\)\s*\{(.*?)\n\}", src, re.S).group(1)
    return [(int(m), int(l)) for m, l in re.findall(r"cmpx<(\d+),\s*(\d+)>", body)]


def _run_network(keys, steps):
    k = list(keys)
    for m, lowbit in steps:                      # cmpx<M, LOWBIT>: lane ^ M is the partner, lanes with (lane & LOWBIT) == 0 keep the minimum
        nk = list(k)
        for lane in range(64):
            o = k[lane ^ m]
            keepmin = (lane & lowbit) == 0
            nk[lane] = k[lane] if ((k[lane] < o) == keepmin) else o
        k = nk
    return k


def test_wave_bitonic_networks_sort():
    """wave_bitonic64 sorts any 64 keys; wave_halfclean64 sorts any bitonic-merged half (checked through
    wave_bitonic128's construction: two sorted 64-runs, element e paired with e ^ 127, then half-cleaners)."""
    rng = np.random.default_rng(3)
    s64 = _network_steps("wave_bitonic64")
    hc = _network_steps("wave_halfclean64")
    assert len(s64) == 21 and len(hc) == 6
    for trial in range(300):
        if trial % 3 == 0:
            keys = rng.integers(0, 2, 64).tolist()                  # 0-1 inputs (zero-one principle, sampled)
        elif trial % 3 == 1:
            keys = rng.integers(0, 5, 64).tolist()                  # many duplicates
        else:
            keys = rng.integers(0, 2**63, 64).tolist()
        assert _run_network(keys, s64) == sorted(keys)
    for trial in range(200):
        keys = rng.integers(0, 50 if trial % 2 else 2**62, 128).tolist()
        k0, k1 = _run_network(keys[:64], s64), _run_network(keys[64:], s64)
        o1, o0 = [k1[l ^ 63] for l in range(64)], [k0[l ^ 63] for l in range(64)]
        k0 = [min(a, b) for a, b in zip(k0, o1)]
        k1 = [max(a, b) for a, b in zip(k1, o0)]
        assert _run_network(k0, hc) + _run_network(k1, hc) == sorted(keys)


def test_match_rank8_bitop3_truth_table():
    """match_rank8: m &= (bit ? v : ~v) per digit bit, as v_bitop3_b32 with truth table 0x90 (index = a*4 + b*2 + c,
    c = the lane's bit replicated): the surviving mask is exactly the lanes with the same digit."""
    tt = 0x90
    bitop3 = lambda a, b, c: sum((((tt >> ((((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1))) & 1) << i) for i in range(64))
    rng = np.random.default_rng(4)
    for _ in range(60):
        digits = rng.integers(0, rng.integers(1, 40), 64)
        live = rng.random(64) > 0.2
        active = sum(1 << l for l in range(64) if live[l])
        for lane in np.nonzero(live)[0][:16]:
            m = active
            for b in range(8):
                beta = -((int(digits[lane]) >> b) & 1) & (2**64 - 1)               # 0 or all ones
                v = sum(1 << l for l in range(64) if live[l] and (int(digits[l]) >> b) & 1)
                m = bitop3(m, v, beta)
            same = [l for l in range(64) if live[l] and digits[l] == digits[lane]]
            assert m == sum(1 << l for l in same)
            assert bin(m & ((1 << int(lane)) - 1)).count("1") == sum(1 for l in same if l < lane)   # stable rank


def test_compensated_atomic_sum_is_order_independent_to_one_ulp():
    """gx_groupby float SUM: every add is a RETURNING atomic, the thread recomputes the rounding error of
    that add from the returned old value (two_sum) and adds it to a compensation word; the result is
    sum + comp.  Modelled here with Python floats: whatever order the adds land in -- also for values of
    wildly different magnitude and sign -- the result stays within 1 ulp of the correctly rounded sum
    (math.fsum), while the plain running sum drifts."""
    import math
    rng = np.random.default_rng(5)

    def model(values):
        s = 0.0
        comp = 0.0
        for v in values:
            old = s
            s = old + v                      # the atomic add; `old` is what ds_add_rtn_f64 / global atomic returns
            bb = s - old                     # two_sum: exact error of old + v
            err = (old - (s - bb)) + (v - bb)
            comp += err                      # second atomic, on the compensation word
        return s + comp, s

    worst_plain = 0
    for case in range(12):
        n = 20000
        if case % 3 == 0:
            v = rng.random(n)
        elif case % 3 == 1:
            v = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)       # 16 orders of magnitude, both signs
        else:
            v = np.concatenate([rng.random(n // 2) * 1e12, -rng.random(n // 2) * 1e12, rng.random(10)])
        exact = math.fsum(v.tolist())
        for _ in range(3):
            order = rng.permutation(len(v))
            got, plain = model(v[order].tolist())
            ulps = abs(got - exact) / math.ulp(exact) if exact != 0 else abs(got)
            assert ulps <= 1.0, (case, ulps)
            worst_plain = max(worst_plain, abs(plain - exact) / math.ulp(exact))
    assert worst_plain > 1.0                 # the uncompensated sum is NOT within 1 ulp on these inputs


def test_decoupled_lookback_protocol_model():
    """The partition passes' tile hand-off (gx_sort.hip k_radix_pass / k_msd_pass): a tile publishes
    {flag 1 = aggregate | 2 = inclusive, epoch, count} in ONE 8-byte granule, walks back over its
    predecessors in windows of LBW granules (loaded as a batch, re-read while a granule is empty or carries
    another epoch), adds aggregates until it meets an inclusive value, then publishes its own inclusive
    value.  The status array is NOT cleared between passes: the epoch makes stale granules invisible.
    Modelled with random interleavings at single load / store granularity, tiles started in ticket order."""
    import random
    LBW = 4

    def tile_program(tile, epoch, count, status, result):
        # publish the aggregate (tile 0 publishes an inclusive value straight away)
        status[tile] = (2 if tile == 0 else 1, epoch, count)
        yield
        prefix = 0
        if tile > 0:
            p, done = tile - 1, False
            while not done:
                window = []
                for k in range(LBW):                       # batch of loads
                    q = p - k
                    window.append(status[q] if q >= 0 else (2, epoch, 0))
                    yield
                for k in range(LBW):
                    if done:
                        break
                    x = window[k]
                    while x[0] == 0 or x[1] != epoch:      # empty or stale: spin on this granule
                        yield
                        x = status[p - k]
                    prefix += x[2]
                    if x[0] == 2:
                        done = True
                p -= LBW
            status[tile] = (2, epoch, prefix + count)
            yield
        result[tile] = prefix

    rnd = random.Random(7)
    for trial in range(60):
        ntiles = rnd.randint(1, 40)
        status = [(0, 0, 0)] * 40                          # reused across epochs, never cleared
        for epoch in range(1, 5):
            counts = [rnd.randint(0, 1000) for _ in range(ntiles)]
            result = [None] * ntiles
            running, started = [], 0
            while started < ntiles or running:
                # tickets: the next tile may start at any time, but only in index order
                if started < ntiles and (not running or rnd.random() < 0.3):
                    running.append(tile_program(started, epoch, counts[started], status, result))
                    started += 1
                    continue
                g = rnd.choice(running)
                try:
                    next(g)
                except StopIteration:
                    running.remove(g)
            assert result == [sum(counts[:t]) for t in range(ntiles)], (trial, epoch)


def test_row_encoding_refinement_algorithm():
    """Multi-column keys (ops.encode_rows / cpp row_encoding.cpp): every column -> dense ids, then
    (ids so far, next column's ids) pairs are packed into 64 bits and ranked again.  Restated with NumPy:
    the final ids are the dense rank of the ROWS in lexicographic order (so sorting by id sorts by key,
    equal rows share an id, null == null as its own last value per column), and the first row of every id
    is the smallest row index of the group."""
    rng = np.random.default_rng(8)

    def dense_rank(values, valid=None):          # what gx_dense_rank returns: ids ascending with the value, nulls last
        n = len(values)
        ok = np.ones(n, bool) if valid is None else valid
        uniq, inv = np.unique(values[ok], return_inverse=True)
        ids = np.full(n, len(uniq), np.int64)
        ids[ok] = inv.reshape(-1)
        return ids

    for trial in range(30):
        n = int(rng.integers(1, 400))
        ncols = int(rng.integers(1, 5))
        cols = [rng.integers(-3, 4, n).astype(rng.choice([np.int8, np.int32, np.int64])) for _ in range(ncols)]
        valids = [rng.random(n) > 0.15 if rng.random() < 0.5 else None for _ in range(ncols)]
        cur = None
        for c, v in zip(cols, valids):
            ids = dense_rank(c, v)
            if cur is None:
                cur = ids
            else:
                pair = (cur.astype(np.uint64) << np.uint64(32)) | ids.astype(np.uint64)      # gx_pack_keys of two int32 ids
                cur = dense_rank(pair)
        # brute force: rows as tuples with null -> (1, 0) after every value (0, v)
        rows = [tuple((1, 0) if (v is not None and not v[i]) else (0, int(c[i])) for c, v in zip(cols, valids)) for i in range(n)]
        order = sorted(set(rows))
        exp = np.array([order.index(r) for r in rows])
        np.testing.assert_array_equal(cur, exp)
        first = {}
        for i, r in enumerate(rows):
            first.setdefault(r, i)
        rep = np.full(len(order), n)
        np.minimum.at(rep, cur, np.arange(n))
        assert [first[r] for r in order] == rep.tolist()


def test_dictionary_row_encoder_model():
    """cpp/src/row_encoding.cpp (cudf::hash_join builds once, probes many): per-column dictionaries learnt
    from the BUILD table, probe values looked up (unseen -> JoinNoMatch = INT32_MIN, null -> the build side's
    null id when nulls compare equal and it has one), (id, id) pairs packed and looked up again.  Restated with
    NumPy and checked against brute-force row equality for inner-join pairs, both null_equality values."""
    rng = np.random.default_rng(9)
    NO = -(2**31)

    def rank(values, valid):
        ok = np.ones(len(values), bool) if valid is None else valid
        uniq = np.unique(values[ok])
        ids = np.full(len(values), len(uniq), np.int64)
        ids[ok] = np.searchsorted(uniq, values[ok])
        return ids, uniq, (len(uniq) if not ok.all() else -1)         # ids, dictionary, null id

    def lookup(uniq, values, valid, null_id, nulls_equal):
        ok = np.ones(len(values), bool) if valid is None else valid
        pos = np.searchsorted(uniq, values)
        hit = ok & (pos < len(uniq)) & (uniq[np.minimum(pos, len(uniq) - 1)] == values) if len(uniq) else np.zeros(len(values), bool)
        ids = np.where(hit, pos, NO).astype(np.int64)
        if nulls_equal and null_id >= 0:
            ids[~ok] = null_id
        return ids

    pack = lambda a, b: ((a.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64) << np.uint64(32)) | (b.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)

    for trial in range(40):
        ncols = int(rng.integers(2, 5))
        nb, npr = int(rng.integers(1, 60)), int(rng.integers(1, 120))
        nulls_equal = bool(trial % 2)
        build = [rng.integers(0, 4, nb) for _ in range(ncols)]
        probe = [rng.integers(-1, 5, npr) for _ in range(ncols)]          # -1 and 4 never occur on the build side
        bval = [rng.random(nb) > 0.2 if rng.random() < 0.6 else None for _ in range(ncols)]
        pval = [rng.random(npr) > 0.2 if rng.random() < 0.6 else None for _ in range(ncols)]
        # ---- build side: dictionaries per column and per inner pair level
        col_dict, pair_dict, cur = [], [], None
        for k in range(ncols):
            ids, uniq, null_id = rank(build[k], bval[k])
            col_dict.append((uniq, null_id))
            if k == 0:
                cur = ids
                continue
            pair = pack(cur, ids)
            if k == ncols - 1:
                bkey = pair
            else:
                pid, puniq, _ = rank(pair, None)
                pair_dict.append(puniq)
                cur = pid
        # ---- probe side
        cur = None
        for k in range(ncols):
            ids = lookup(col_dict[k][0], probe[k], pval[k], col_dict[k][1], nulls_equal)
            if k == 0:
                cur = ids
                continue
            pair = pack(cur, ids)
            if k == ncols - 1:
                pkey = pair
            else:
                cur = lookup(pair_dict[k - 1], pair, None, -1, nulls_equal)
        b_ok = np.ones(nb, bool)
        p_ok = np.ones(npr, bool)
        if not nulls_equal:                                               # rows holding a null match nothing
            for v in bval:
                if v is not None:
                    b_ok &= v
            for v in pval:
                if v is not None:
                    p_ok &= v
        got = sorted((i, j) for i in range(npr) for j in range(nb) if p_ok[i] and b_ok[j] and pkey[i] == bkey[j])

        def row(cols, vals, i):
            return tuple(None if (v is not None and not v[i]) else int(c[i]) for c, v in zip(cols, vals))
        exp = []
        for i in range(npr):
            ri = row(probe, pval, i)
            for j in range(nb):
                rj = row(build, bval, j)
                if ri == rj and (nulls_equal or None not in ri):
                    exp.append((i, j))
        assert got == sorted(exp), (trial, nulls_equal)


def test_xcd_lists_with_stealing_make_progress_with_few_resident_workgroups():
    """Hybrid sort levels / partitioned join: tiles are grouped in 8 lists (one per XCD), every list has its own
    ticket counter and its own look-back chain; a workgroup takes a ticket from the list of the XCD it runs on
    and steals from the others when its own is exhausted.  Placement is only a performance hint: because a tile's
    predecessors in its chain always hold EARLIER tickets of the same list, they are already running (or done)
    when the tile starts, so the chain cannot wait on a tile nobody has started -- even when only a handful of
    workgroups are resident.  Modelled with W resident slots, random XCD placement and random interleaving."""
    import random
    rnd = random.Random(11)

    def worker(xcd, tickets, lists, status, result, counts):
        # take a tile: own list first, then the others (pj_xcc() + i) % 8
        tile = None
        for i in range(8):
            y = (xcd + i) % 8
            if tickets[y] < len(lists[y]):
                tile = (y, tickets[y])
                tickets[y] += 1
                break
        if tile is None:
            return
        y, t = tile
        yield
        status[y][t] = (2 if t == 0 else 1, counts[y][t])
        yield
        prefix = 0
        p = t - 1
        while t > 0:
            x = status[y][p]
            while x[0] == 0:                    # predecessor has not published yet: spin (it IS running)
                yield
                x = status[y][p]
            prefix += x[1]
            if x[0] == 2:
                break
            p -= 1
        if t > 0:
            status[y][t] = (2, prefix + counts[y][t])
        result[y][t] = prefix
        yield

    for trial in range(40):
        W = rnd.randint(1, 6)                   # resident workgroups (far fewer than tiles)
        lists = [list(range(rnd.randint(0, 12))) for _ in range(8)]
        counts = [[rnd.randint(0, 99) for _ in l] for l in lists]
        status = [[(0, 0)] * len(l) for l in lists]
        result = [[None] * len(l) for l in lists]
        tickets = [0] * 8
        total = sum(len(l) for l in lists)
        launched, running, steps = 0, [], 0
        while launched < total + W or running:  # the grid has a few spare workgroups that find nothing to do
            while len(running) < W and launched < total + W:
                running.append(worker(rnd.randrange(8), tickets, lists, status, result, counts))
                launched += 1
            g = rnd.choice(running)
            try:
                next(g)
            except StopIteration:
                running.remove(g)
            steps += 1
            assert steps < 200000, "no progress"
        for y in range(8):
            assert result[y] == [sum(counts[y][:t]) for t in range(len(lists[y]))]


def test_hashed_row_keys_certificates_detect_every_collision_and_nothing_else():
    """The multi-column-key operators key rows by a 64-bit hash and CERTIFY the result.  Model of the
    two certificates with a deliberately weak hash (so that collisions happen): equal rows hash equal, hence a
    hashed join never misses a pair and a hashed groupby never splits a group; the join certificate (compare every
    emitted pair) and the groupby certificate (MIN == MAX of every key column inside every hash group) fire exactly
    when two DIFFERENT rows share a hash, and when they stay silent the hashed result equals the exact one."""
    rng = np.random.default_rng(5)

    def weak_hash(cols, bits):
        h = np.zeros(len(cols[0]), np.uint64)
        for c in cols:
            h = (h * np.uint64(1000003) + c.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64((1 << 63) - 1)
        return h >> np.uint64(63 - bits)

    saw_collision = saw_clean = False
    for trial in range(60):
        bits = int(rng.integers(3, 14))
        n, m = int(rng.integers(1, 300)), int(rng.integers(1, 80))
        card = int(rng.integers(2, 12))
        L = [rng.integers(0, card, n), rng.integers(0, card, n)]
        R = [rng.integers(0, card, m), rng.integers(0, card, m)]
        # ---- join
        hl, hr = weak_hash(L, bits), weak_hash(R, bits)
        hashed = {(i, j) for i in range(n) for j in range(m) if hl[i] == hr[j]}
        exact = {(i, j) for i in range(n) for j in range(m) if L[0][i] == R[0][j] and L[1][i] == R[1][j]}
        assert exact <= hashed                                   # no pair can be missing
        mismatches = sum(1 for (i, j) in hashed if not (L[0][i] == R[0][j] and L[1][i] == R[1][j]))
        assert (mismatches == 0) == (hashed == exact)            # certificate silent <=> result exact
        # ---- groupby
        groups = {}
        for i in range(n):
            groups.setdefault(int(hl[i]), []).append(i)
        cert_ok = all(min(c[i] for i in g) == max(c[i] for i in g) for g in groups.values() for c in L)
        exact_groups = {}
        for i in range(n):
            exact_groups.setdefault((int(L[0][i]), int(L[1][i])), []).append(i)
        same_partition = sorted(map(tuple, groups.values())) == sorted(map(tuple, exact_groups.values()))
        assert cert_ok == same_partition
        if cert_ok:   # the MINs are the group keys
            assert sorted((min(L[0][i] for i in g), min(L[1][i] for i in g)) for g in groups.values()) == sorted(exact_groups)
        saw_collision |= not cert_ok or mismatches > 0
        saw_clean |= cert_ok and mismatches == 0
    assert saw_collision and saw_clean


def test_single_pass_lookback_scan_protocol_model():
    """gx_scan's single-pass look-back (gx_scan.hpp k_lookback_scan): tiles taken by ticket publish {flag 1, aggregate},
    fold the LB_WIN nearest predecessors per round up to the nearest {flag 2, inclusive prefix} and publish their own.
    An accumulator travels as two granules that are written one after the other; a reader accepts a pair only when
    both carry the same non-zero flag.  Random interleavings of (publish lo, publish hi, read) steps must always give
    the sequential exclusive prefix, whatever the order tiles make progress in."""
    rng = np.random.default_rng(9)
    WIN = 4
    for trial in range(200):
        ntiles = int(rng.integers(1, 24))
        agg = rng.integers(-1000, 1000, ntiles)
        want = np.concatenate([[0], np.cumsum(agg)[:-1]])
        lo = [(0, 0)] * ntiles      # (flag, low half)
        hi = [(0, 0)] * ntiles
        # per-tile program counters: 0/1 publish aggregate (lo, hi); 2.. look back; final publish prefix (lo, hi)
        state = [{"pc": 0, "pos": t - 1, "ex": 0, "done": False, "pend": None} for t in range(ntiles)]
        started = 0                 # tickets: tile t may only start once tiles < t have started
        got = [None] * ntiles
        guard = 0
        while not all(s["done"] for s in state):
            guard += 1
            assert guard < 100000
            runnable = [t for t in range(min(ntiles, started + 1)) if not state[t]["done"]]
            t = int(rng.choice(runnable))
            started = max(started, t + 1)
            s = state[t]
            first_flag = 2 if t == 0 else 1
            if s["pc"] == 0:
                lo[t] = (first_flag, int(agg[t]) & 0xFFFF); s["pc"] = 1
            elif s["pc"] == 1:
                hi[t] = (first_flag, int(agg[t]) >> 16); s["pc"] = 2
                if t == 0:
                    got[0] = 0; s["done"] = True
            elif s["pc"] == 2:      # one look-back round (all-or-nothing: a lane spins until its word is consistent)
                window = [s["pos"] - k for k in range(WIN) if s["pos"] - k >= 0]
                vals, ok = [], True
                for q in window:
                    (fl, l), (fh, h) = lo[q], hi[q]
                    if fl == 0 or fl != fh:
                        ok = False; break
                    vals.append((fl, (h << 16) | l))
                if not ok:
                    continue        # spin: try again later
                done = False
                for fl, v in vals:
                    v = v - (1 << 32) if v >= (1 << 31) else v
                    s["ex"] += v
                    if fl == 2:
                        done = True; break
                if not window:
                    done = True
                if done:
                    s["pc"] = 3
                else:
                    s["pos"] -= WIN
                    if s["pos"] < 0:
                        s["pc"] = 3
            elif s["pc"] == 3:
                lo[t] = (2, int(s["ex"] + agg[t]) & 0xFFFF); s["pc"] = 4
            else:
                hi[t] = (2, int(s["ex"] + agg[t]) >> 16)
                got[t] = s["ex"]; s["done"] = True
        assert got == [int(x) for x in want], (trial, got, want.tolist())


def _ce_list(name):
    """The compare-exchange list of a per-thread register network, parsed from gx_common.hpp."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cudf_amd", "csrc", "gx_common.hpp")).read()
    body = re.search(r"void " + name + r"\(T \(&v\)\[16\]\)\s*\{(.*?)\n\}", src, re.S).group(1)
    return [(int(a), int(b)) for a, b in re.findall(r"GX_CE\((\d+),\s*(\d+)\)", body)]


def test_register_networks_of_the_placement_sort():
    """k_local_place (gx_sort.hip): sort16_regs must sort every input (0-1 principle over all 2^16 inputs), merge16_regs must
    merge any two ascending runs of 8; and the kernel's claim -- keys scattered to 13-bit bins of <= 9 keys each are sorted
    by one pass over aligned 16-key windows and one over windows shifted by 8 -- on a model of a cell."""
    s16, m16 = _ce_list("sort16_regs"), _ce_list("merge16_regs")
    assert len(s16) == 60 and len(m16) == 25
    bits = ((np.arange(1 << 16)[:, None] >> np.arange(16)[None, :]) & 1).astype(np.uint8)
    for a, b in s16:
        lo, hi = np.minimum(bits[:, a], bits[:, b]), np.maximum(bits[:, a], bits[:, b])
        bits[:, a], bits[:, b] = lo, hi
    assert (np.diff(bits.astype(np.int8), axis=1) >= 0).all()
    runs = np.array([[0] * za + [1] * (8 - za) + [0] * zb + [1] * (8 - zb) for za in range(9) for zb in range(9)], dtype=np.uint8)
    for a, b in m16:
        lo, hi = np.minimum(runs[:, a], runs[:, b]), np.maximum(runs[:, a], runs[:, b])
        runs[:, a], runs[:, b] = lo, hi
    assert (np.diff(runs.astype(np.int8), axis=1) >= 0).all()

    rng = np.random.default_rng(0)

    def windows(keys, shift2):
        m = len(keys)
        bins = (keys >> np.uint64(shift2 - 13)).astype(np.int64)
        assert np.bincount(bins, minlength=8192).max() <= 9
        perm = rng.permutation(m)  # arrival order inside a bin is whatever the atomics return
        cell = np.full(8192 + 16, np.uint64(2**64 - 1))
        cell[:m] = keys[perm[np.argsort(bins[perm], kind="stable")]]
        cell[:8192].reshape(512, 16).sort(axis=1)
        cell[8:8200].reshape(512, 16).sort(axis=1)
        return cell[:m]

    for m in (1, 17, 4000, 7629, 8192):
        keys = rng.integers(0, 1 << 47, m, dtype=np.uint64)
        while np.bincount((keys >> np.uint64(34)).astype(np.int64), minlength=8192).max() > 9:
            keys = rng.integers(0, 1 << 47, m, dtype=np.uint64)
        np.testing.assert_array_equal(windows(keys, 47), np.sort(keys))
    # every bin exactly 9 keys, bins straddling every window boundary
    keys = (np.repeat(np.arange(910, dtype=np.uint64), 9) << np.uint64(34)) | rng.integers(0, 1 << 34, 8190, dtype=np.uint64)
    np.testing.assert_array_equal(windows(keys, 47), np.sort(keys))
    # the byte-counter arithmetic of the kernel: x * 0x01010101 is the inclusive prefix of the four bytes of x while sums < 256
    c = rng.integers(0, 10, (1000, 4), dtype=np.uint32)
    x = c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16) | (c[:, 3] << 24)
    p = (x.astype(np.uint64) * 0x01010101) & 0xFFFFFFFF
    inc = np.cumsum(c, axis=1)
    for k in range(4):
        np.testing.assert_array_equal((p >> (8 * k)) & 0xFF, inc[:, k])
    ge10 = ((((x & 0x7F7F7F7F) + 0x76767676) | x) & 0x80808080) != 0
    assert not ge10.any()
    y = x.copy(); y[::3] |= np.uint32(10) << (8 * rng.integers(0, 4, len(y[::3]), dtype=np.uint32))
    ge10 = ((((y & 0x7F7F7F7F) + 0x76767676) | y) & 0x80808080) != 0
    np.testing.assert_array_equal(ge10, ((y[:, None] >> (8 * np.arange(4, dtype=np.uint32))) & 0xFF).max(axis=1) >= 10)


def test_pmc_kernel_names_and_groups():
    """scripts/pmc_to_json.py: rocprofv3 kernel names -> the base names bench.py's `traffic_key`s refer to"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pmc_to_json", os.path.join(root, "scripts", "pmc_to_json.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bn = mod.base_name
    assert bn("void gx::sort::k_hf_scatter<unsigned long, 1, 0, 8>(unsigned long const*, unsigned long*)") == "k_hf_scatter level 0"
    assert bn("void gx::sort::k_hf_scatter<unsigned long, 1, 1, 10>(unsigned long const*)") == "k_hf_scatter level 1"
    assert bn("void gx::sort::k_local_place<unsigned long, 1, false, 13>(unsigned long const*)") == "k_local_place"
    assert bn("void gx::sort::k_msd_pass<unsigned long, 1, true, 10, 4, 9>(gx::sort::MsdArgs)") == "k_msd_pass level 1"
    assert bn("void gx::sort::k_msd_pass<unsigned long, 1, true, 10, 4, 8>(gx::sort::MsdArgs)") == "k_msd_pass level 0"
    assert bn("gx::join::k_pj2_offsets(gx::join::Pj2Plan*, int, unsigned int, unsigned int)") == "k_pj2_offsets"
    assert bn("void at::native::vectorized_elementwise_kernel<4>(int)") is None
    # every traffic key bench.py uses resolves in the committed round-3 file
    import json
    tr = json.load(open(os.path.join(root, "profiles", "r3_pmc_traffic_1e9.json")))
    for key in ("k_hf_scatter level 0", "k_hf_scatter level 1", "sort local stage", "join probe phase", "groupby"):
        e = tr["groups"].get(key) or tr["kernels"].get(key)
        assert e and e["hbm_bytes_per_launch"] > 1e9, key
    assert 48e9 < tr["groups"]["sort"]["hbm_bytes_per_launch"] < 51e9


def _sortable_i64(v, descending=False):
    s = v.view(np.uint64) ^ np.uint64(1 << 63)
    return ~s if descending else s


def test_sign_fold_digit_and_lsd_skip_rule_models():
    """gx_sort.hip round 4, two plan rules for signed keys spread around zero, restated in NumPy on the kernels' own bit formulas.
    (1) fold_height / Digit0: with h = bit length of OR(key ^ sign extension), sorting by ((sign << 7) | bits [h-7, h) of the
    sortable key, then the bits below h - 7) is sorting by the key -- the bits in [h, 63) are copies of the sign, equal inside a
    bucket.  (2) k_plan: stable LSD passes over the bytes below h plus the TOP byte order the column; the bytes wholly inside
    [h, 56) are skipped.  Both for ascending and descending sortable forms and for ranges on one side of zero."""
    rng = np.random.default_rng(4)
    for lo, hi in [(-10**12, 10**12), (-1000, 1000), (-(1 << 40), 1 << 33), (-5, 3 * 10**15), (0, 1000), (-3000, -1000), (-1, 1)]:
        v = rng.integers(lo, hi, 20000, dtype=np.int64)
        v[:2] = lo, hi - 1
        fold_or = int(np.bitwise_or.reduce((v ^ (v >> 63)).view(np.uint64)))
        h = fold_or.bit_length()
        for desc in (False, True):
            s = _sortable_i64(v, desc)
            want = np.sort(v)[::-1] if desc else np.sort(v)
            # (2) the LSD plan: passes on byte p unless it is constant or (p < 7 and 8p >= h)
            order = np.arange(len(v))
            passes = 0
            for p in range(8):
                d = ((s >> np.uint64(8 * p)) & np.uint64(0xFF)).astype(np.int64)
                if len(np.unique(d)) == 1 or (p < 7 and 8 * p >= h):
                    continue
                order = order[np.argsort(d[order], kind="stable")]
                passes += 1
            np.testing.assert_array_equal(v[order], want)
            assert passes <= (h + 7) // 8 + 1
            # (1) the folded level-0 digit (planned only when the sign varies and 7 <= h <= 54)
            sign_varies = (v < 0).any() and (v >= 0).any()
            if sign_varies and 7 <= h <= 54:
                d0 = ((s >> np.uint64(h - 7)) & np.uint64(0x7F)) | ((s >> np.uint64(56)) & np.uint64(0x80))
                low = s & np.uint64((1 << (h - 7)) - 1)
                order = np.lexsort((low, d0))
                np.testing.assert_array_equal(v[order], want)
                # every bucket's keys agree on every bit from h - 7 upwards: what the cell sort below relies on
                top = s >> np.uint64(h - 7)
                for b in np.unique(d0)[:8]:
                    assert len(np.unique(top[d0 == b])) == 1


def test_early_decline_estimate_is_a_lower_bound_times_two_at_most():
    """k_hf_plan stage 2: keys of a bucket that must sit in overflowing cells.  With K cells of `cap` keys a bucket of c keys has at
    least c - (K - 1) * cap keys in cells above capacity (everything else filled to the brim); the kernel counts the bucket whole
    once c > 2 * K * cap, which overstates that bound by less than 2x.  Checked against a brute-force worst case."""
    K, cap = 8, 100
    for c in [0, 500, 800, 801, 950, 1600, 1601, 5000]:
        fits, full = (K - 1) * cap, K * cap
        over = c if c > 2 * full else (c - fits if c > full else 0)
        # fewest keys in overflowing cells: fill K - 1 cells to exactly cap, the rest into the last one
        rest = c - fits
        brute = rest if rest > cap else 0
        assert brute <= over or c <= full
        if c > 2 * full:
            assert over < 2 * brute


def test_splitter_sort_model_orders_every_distribution():
    """DESIGN.md section 8 row 1 (proposed, not built): the maps of the splitter sort are MONOTONE, so concatenating the sorted
    cells in (bucket, cell) order is the sorted column -- executable statement of the plan next round's kernels have to keep:
    bucket = upper_bound(splitters, key) with an equality bucket [v, v + 1) for a value that fills two sample quantiles;
    cell = floor(rel * ncell / width) on rel = key - bucket_lo, exactly as a 64 x 64 -> 128-bit multiply-high computes it."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("xp_splitter_model", os.path.join(os.path.dirname(__file__), "..", "scripts", "xp", "xp_splitter_model.py"))
    xp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(xp)
    rng = np.random.default_rng(3)
    n = 1 << 17
    for name, s in xp.distributions(n, rng):
        samp = np.sort(s[rng.integers(0, n, 4096)])
        q = samp[(np.arange(1, 256) * 4096) // 256]
        vals, reps = np.unique(q, return_counts=True)
        sp = np.unique(np.concatenate([vals, vals[reps >= 2] + np.uint64(1)]))
        bucket = np.searchsorted(sp, s, side="right")
        lo = np.concatenate([[s.min()], sp])
        hi = np.concatenate([sp, [s.max() + np.uint64(1)]])
        out = []
        for b in range(len(sp) + 1):
            kb = s[bucket == b]
            if len(kb) == 0:
                continue
            width = int(hi[b]) - int(lo[b])
            ncell = max(1, min(-(-len(kb) // 512), 1024, width))            # (512-key cells here: several cells per bucket at this n)
            rel = [int(k) - int(lo[b]) for k in kb]
            cell = np.array([(r * ncell) // width for r in rel])              # exact integer arithmetic = the multiply-high
            assert cell.min() >= 0 and cell.max() < ncell, name
            for k in range(ncell):
                out.append(np.sort(kb[cell == k]))
        got = np.concatenate(out)
        np.testing.assert_array_equal(got, np.sort(s), err_msg=name)


# ------------------------------------------------------------------------------------------------
# Round 5: the fraction map of the sort's splitter mode (gx_sort.hip: k_sp_plan's per-bucket parameters, sp_frac, sp_cell, sp_fine)
# ------------------------------------------------------------------------------------------------
def _sp_params(w):
    """what k_sp_plan stores for a bucket of `w` key values: (nsh, mlow) -- the range's last key, relative, becomes a 31-bit number with
    bit 30 set; M = floor(2^63 / (that + 1)) lies in [2^32, 2^33) and mlow = M - 2^32"""
    wm = w - 1
    if wm == 0:
        return 63, 0
    wl = wm.bit_length()
    nsh = wl - 31
    wx = wm >> nsh if nsh >= 0 else wm << -nsh
    assert (wx >> 30) == 1
    M = (1 << 63) // (wx + 1)
    assert (1 << 32) <= M < (1 << 33)
    return nsh, M - (1 << 32)


def _sp_frac(rel, w, nsh, mlow):
    """sp_frac: position of key = lo + rel inside [lo, lo + w) as a 32-bit fraction (the device's integer arithmetic, width for width)"""
    if rel >= w:
        return 0xFFFFFFFF
    x = (rel >> nsh) if nsh >= 0 else ((rel << -nsh) & 0xFFFFFFFF)
    x &= 0xFFFFFFFF
    p = x * mlow
    f = ((x << 1) & 0xFFFFFFFF) + ((p >> 31) & 0xFFFFFFFF)
    assert f < (1 << 32), "the sum must not wrap (x * M / 2^31 < 2^32)"
    return f


def test_splitter_fraction_map_is_monotone_and_bounded():
    """two unordered partition levels and a cell sort on full keys need nothing of the cell map but MONOTONICITY (a key never lands in a
    cell before a smaller key's) and cell < cells: checked on bucket widths from 1 to 2^64 - 1 and cell counts up to 1024"""
    import random
    rnd = random.Random(7)
    widths = [1, 2, 3, 5, 1023, 1024, 1025, 65535, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 40) + 12345, (1 << 63) + 9, (1 << 64) - 1]
    widths += [rnd.getrandbits(rnd.randrange(1, 65)) | 1 for _ in range(60)]
    for w in widths:
        nsh, mlow = _sp_params(w)
        rels = sorted({0, w - 1, w // 2, w // 3} | {rnd.randrange(w) for _ in range(300)} | {min(w - 1, k) for k in range(40)})
        fr = [_sp_frac(r, w, nsh, mlow) for r in rels]
        assert all(a <= b for a, b in zip(fr, fr[1:])), w
        assert _sp_frac(w, w, nsh, mlow) == 0xFFFFFFFF and fr[-1] <= 0xFFFFFFFF   # keys past the range clamp to its end
        for nc in (1, 7, 512, 700, 1024):
            cells = [(f * nc) >> 32 for f in fr]
            assert all(a <= b for a, b in zip(cells, cells[1:])) and cells[-1] < nc
            fine = [((f * nc) >> 19) for f in fr]                                   # cell * 8192 + the 13-bit counting digit
            assert all(a <= b for a, b in zip(fine, fine[1:]))


def test_splitter_narrow_bucket_has_one_value_per_cell():
    """a NARROW bucket (no more values in its range than it has cell slots) is counted and filled instead of sorted: that needs the
    cell map to be INJECTIVE on the bucket's values -- k_sp_fill inverts it by evaluating it on every value"""
    for ncmax in (64, 256, 1024):
        for w in list(range(2, 70)) + [ncmax - 1, ncmax, ncmax // 2 + 1, ncmax * 3 // 4]:
            if w > ncmax:
                continue
            nsh, mlow = _sp_params(w)
            cells = [(_sp_frac(r, w, nsh, mlow) * ncmax) >> 32 for r in range(w)]
            assert len(set(cells)) == w and cells == sorted(cells) and cells[-1] < ncmax, (ncmax, w)


def _sp_warp_table(masses):
    """k_hf_plan stage 1 (splitter mode): the sixteen (start, stretch) pairs of a bucket's warp from the sampled masses of its pieces"""
    M = sum(masses)
    fl = M // (16 * 64) + 1
    m = [max(x, fl) for x in masses]
    tot = float(sum(m))
    y, tab = 0, []
    for x in m:
        d = max(1, int(float(x) / tot * 4294967040.0))
        tab.append((y, d))
        y += d
    assert y < (1 << 32)
    return tab


def _sp_warp(frac, tab):
    y0, d = tab[frac >> 28]
    return y0 + (((frac & 0x0FFFFFFF) * d) >> 28)


def test_splitter_warp_is_monotone_bounded_and_the_identity_for_narrow_buckets():
    """sp_warp (gx_sort.hip): cells are equal slices of the WARPED fraction -- a piecewise-linear estimate of the bucket's CDF from the
    n / 32 sample.  The levels need it monotone and below 2^32 for every table stage 1 can produce (any masses, zeros included); narrow
    buckets need the identity table to reproduce the fraction bit for bit (k_sp_fill inverts the un-warped cell map)."""
    import random
    rnd = random.Random(11)
    ident = [(j << 28, 1 << 28) for j in range(16)]
    fracs = sorted({0, 1, 0x0FFFFFFF, 0x10000000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF} | {rnd.getrandbits(32) for _ in range(2000)}
                   | {(j << 28) + k for j in range(16) for k in (0, 1, 0x0FFFFFFF)})
    assert all(_sp_warp(f, ident) == f for f in fracs)
    shapes = [[7600] * 16, [0] * 15 + [100000], [100000] + [0] * 15, [1 << 27] * 16, [1] * 16, [0, 0, 5, 900000, 5, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 1],
              [int(7600 * 1.3 ** j) for j in range(16)], [int(7600 * 0.5 ** j) + 1 for j in range(16)]]
    shapes += [[rnd.randrange(0, 1 << rnd.randrange(1, 28)) for _ in range(16)] for _ in range(40)]
    for masses in shapes:
        if sum(masses) == 0:
            continue
        tab = _sp_warp_table(masses)
        assert all(tab[j][0] + tab[j][1] <= tab[j + 1][0] for j in range(15))          # piece j ends where piece j + 1 starts, or below
        ys = [_sp_warp(f, tab) for f in fracs]
        assert all(a <= b for a, b in zip(ys, ys[1:])), masses
        assert ys[-1] < (1 << 32)
        for nc in (1, 7, 539, 1024):
            cells = [(y * nc) >> 32 for y in ys]
            assert all(a <= b for a, b in zip(cells, cells[1:])) and cells[-1] < nc
    # what it is for: a bucket whose density doubles from one end to the other -- cells cut on the warped fraction hold equal shares
    masses = [int(100000 * (1 + j / 15.0)) for j in range(16)]
    tab = _sp_warp_table(masses)
    nc = 40
    keys = [rnd.random() for _ in range(200000)]
    keys = [((1 + 3 * u) ** 0.5 - 1) for u in keys]   # density ~ (1 + 2x)/2 on [0, 1): CDF (x + x^2) / 2 ... inverse of u = (x + x^2)/2 scaled
    keys = [min(k, 0.999999999) for k in keys]
    cnt = [0] * nc
    for k in keys:
        cnt[(_sp_warp(int(k * (1 << 32)), tab) * nc) >> 32] += 1
    mean = len(keys) / nc
    assert max(cnt) < 1.06 * mean and min(cnt) > 0.94 * mean, (max(cnt) / mean, min(cnt) / mean)


# ------------------------------------------------------------------------------------------------
# Round 5: the splitter search of level 0 (gx_sort.hip: sp_lut_cell / sp_lut_cell_k, the LUT word, the branch-free steps) and
# k_sp_plan's gap splitters
# ------------------------------------------------------------------------------------------------
_NLUT = 4096


def _lut_cell(key, form, kmin, lshift, f):
    """LUT cell of a sortable 64-bit key: form 0 linear in key - kmin, 1 logarithmic (exponent + 6 mantissa bits), 2 two linear halves
    cut at f = (nmin, pmin, nsh, psh)"""
    if form == 2:
        nmin, pmin, nsh, psh = f
        pos = key >= pmin
        o = pmin if pos else nmin
        c = ((key - o) if key > o else 0) >> (psh if pos else nsh)
        return (_NLUT // 2 if pos else 0) + min(c, _NLUT // 2 - 1)
    rel = key - kmin if key >= kmin else 0
    if form == 1:
        if rel == 0:
            return 0
        e = rel.bit_length() - 1
        m = (rel >> (e - 6)) & 63 if e >= 6 else (rel << (6 - e)) & 63
        return e * 64 + m
    return min(rel >> lshift, _NLUT - 1)


def _build_lut(tab, form, kmin, lshift, f):
    """k_sp_plan: per LUT cell c the splitters in cells below c (bits 0-8) | the splitters inside it, capped at 63 (bits 9-14) | 0x8000
    when it holds none; and the fullest cell's count (SplitPlan::lut_steps)"""
    cells = [_lut_cell(t, form, kmin, lshift, f) for t in tab]
    assert cells == sorted(cells), "the cell function must be monotone over the sorted table"
    lut, worst = [], 0
    import bisect
    for c in range(_NLUT):
        a = bisect.bisect_left(cells, c)
        ins = bisect.bisect_right(cells, c) - a
        worst = max(worst, ins)
        lut.append(a | (min(ins, 63) << 9) | (0x8000 if ins == 0 else 0))
    return lut, worst


def _bucket_steps(tab, lut, key, steps, form, kmin, lshift, f):
    """k_sp_level0's branch-free search: start at the LUT word's count, then `steps` times one further if that splitter is <= key"""
    b = lut[_lut_cell(key, form, kmin, lshift, f)] & 0x1FF
    for _ in range(steps):
        if b < len(tab) and tab[b] <= key:
            b += 1
    return b


def test_splitter_search_by_lut_and_fixed_steps_is_upper_bound_for_every_lut_form():
    """bucket(key) = number of splitters <= key, for keys inside, between, below and above the sampled range, under all three LUT forms --
    which needs nothing of a form but that its cell function is monotone; the fixed number of steps is the fullest cell's count"""
    import bisect
    import random
    rnd = random.Random(23)
    shapes = {
        "bell": sorted({(1 << 63) + int(rnd.gauss(0, 2.0 ** 40)) for _ in range(250)}),
        "power law": sorted({int((1.0 - rnd.random()) ** -5.0) + 1000 for _ in range(250)}),
        "two clusters": sorted({(1 << 63) + s * (1 << 50) + int(rnd.gauss(0, 2.0 ** 30)) for _ in range(125) for s in (-1, 1)}),
        "float64 N(0, 1)": None,
    }
    import struct
    def fkey(x):
        b = struct.unpack("<Q", struct.pack("<d", x))[0]
        return (~b) & ((1 << 64) - 1) if b >> 63 else b | (1 << 63)
    shapes["float64 N(0, 1)"] = sorted({fkey(rnd.gauss(0, 1)) for _ in range(250)})
    for name, tab in shapes.items():
        kmin = tab[0] - (tab[1] - tab[0]) // 3
        kmax = tab[-1] + 5
        lshift = 0
        while ((kmax - kmin) >> lshift) >= _NLUT:
            lshift += 1
        gaps = [tab[i + 1] - tab[i] for i in range(len(tab) - 1)]
        gi = gaps.index(max(gaps))
        alast = max(tab[gi], kmin)
        nsh = psh = 0
        while ((alast - kmin) >> nsh) >= _NLUT // 2:
            nsh += 1
        while kmax > tab[gi + 1] and ((kmax - tab[gi + 1]) >> psh) >= _NLUT // 2:
            psh += 1
        f = (kmin, tab[gi + 1], nsh, psh)
        keys = set(tab) | {t - 1 for t in tab} | {t + 1 for t in tab} | {0, 1, kmin, kmin - 1, kmax, (1 << 64) - 1, tab[gi] + gaps[gi] // 2}
        keys |= {rnd.randrange(tab[0], tab[-1]) for _ in range(2000)}
        worsts = []
        for form in (0, 1, 2):
            lut, worst = _build_lut(tab, form, kmin, lshift, f)
            worsts.append(worst)
            for k in keys:
                if k < 0:
                    continue
                assert _bucket_steps(tab, lut, k, worst, form, kmin, lshift, f) == bisect.bisect_right(tab, k), (name, form, k)
                assert _bucket_steps(tab, lut, k, worst + 2, form, kmin, lshift, f) == bisect.bisect_right(tab, k)   # extra steps change nothing
        # what the forms are for: the form k_sp_plan picks (fewest splitters in the fullest cell) needs at most a handful of steps
        assert min(worsts) <= 4, (name, worsts)
        if name in ("two clusters", "float64 N(0, 1)"):
            assert worsts[2] < worsts[0] and worsts[2] < worsts[1], (name, worsts)   # one linear half per cluster / per sign


def test_gap_splitters_keep_the_table_sorted_and_isolate_the_gap():
    """k_sp_plan's gap splitters: a bucket whose sampled keys leave more than half of its extent empty in one gap gets two more
    splitters -- one past the key below the gap, one at the key above it -- so the gap becomes a bucket of its own"""
    import bisect
    samp = sorted([100 + i for i in range(40)] + [10 ** 9 + 3 * i for i in range(40)])      # one bucket's sampled keys: two clusters
    tab = [50, 2 * 10 ** 9]                                                                   # the bucket is [50, 2e9)
    ia, ib = bisect.bisect_left(samp, tab[0]), bisect.bisect_left(samp, tab[1])
    assert ib - ia >= 24
    gaps = [(samp[i + 1] - samp[i], i) for i in range(ia, ib - 1)]
    g, i = max(gaps)
    assert g >= (1 << 20) and g > (samp[ib - 1] - samp[ia]) // 2
    v1, v2 = samp[i] + 1, samp[i + 1]
    new = tab[:1] + [v1, v2] + tab[1:]
    assert new == sorted(new) and len(set(new)) == len(new)
    # every sampled key of the lower cluster is in [50, v1), every one of the upper in [v2, 2e9), none in the gap bucket [v1, v2)
    b = [bisect.bisect_right(new, k) for k in samp]
    assert set(b[:40]) == {1} and set(b[40:]) == {3}
    # a smooth bucket (68 evenly spread keys) never qualifies: its widest gap is a few per cent of its extent
    smooth = sorted({1000 + 977 * i * i % 100000 for i in range(68)})
    gs = max(smooth[i + 1] - smooth[i] for i in range(len(smooth) - 1))
    assert gs <= (smooth[-1] - smooth[0]) // 2
