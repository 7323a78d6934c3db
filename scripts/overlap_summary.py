"""From a rocprofv3 --kernel-trace results.db: per queue / stream the window its kernels ran in, and the time during which kernels of
TWO queues were running at once (scripts/xp/xp_two_streams.cpp: two cudf::sort calls on two streams).  usage: overlap_summary.py <db> [title]"""
import sqlite3
import sys


def main(db, title=""):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    key = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
    print(f"# {title}")
    print(f"# kernels view columns: {cols}; grouping by: {key}")
    if key is None:
        return
    rows = cur.execute(f"select {key}, name, start, end from kernels order by start").fetchall()
    t0 = rows[0][2]
    by = {}
    for q, name, s, e in rows:
        by.setdefault(q, []).append((s - t0, e - t0, name))
    for q, ks in by.items():
        busy = sum(e - s for s, e, _ in ks)
        print(f"{key} {q}: {len(ks)} kernels, first start {ks[0][0] / 1e6:.3f} ms, last end {max(e for _, e, _ in ks) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms")
    # time with kernels of >= 2 different queues in flight (sweep over the start / end points)
    ev = []
    for q, ks in by.items():
        for s, e, _ in ks:
            ev.append((s, 1, q))
            ev.append((e, -1, q))
    ev.sort()
    live, last, both, anyt = {}, 0, 0, 0
    for t, d, q in ev:
        nq = sum(1 for v in live.values() if v > 0)
        if nq >= 1:
            anyt += t - last
        if nq >= 2:
            both += t - last
        live[q] = live.get(q, 0) + d
        last = t
    print(f"time with a kernel running: {anyt / 1e6:.3f} ms; with kernels of two or more {key}s running at once: {both / 1e6:.3f} ms ({100.0 * both / max(anyt, 1):.0f} %)")
    # the big kernels, interleaved as they ran
    print("# kernels longer than 0.2 ms, in start order: start ms | end ms | " + key + " | name")
    for q, name, s, e in rows:
        if e - s > 200_000:
            print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f}  {q}  {name[:90]}")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
