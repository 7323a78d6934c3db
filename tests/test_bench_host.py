"""CPU: host logic of bench.py that needs no device -- the order in which Ctx.timed drives the event slots of the kernels' HIP events
(step k records into slot k % 64; every slot is read AFTER the closing barrier: no read-back between two timed steps)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(steps, warmup):
    sys.path.insert(0, ROOT)
    import bench

    class A:
        pass
    a = A()
    a.steps, a.warmup = steps, warmup
    c = bench.Ctx.__new__(bench.Ctx)
    c.args, c.world, c.beat = a, 1, 0.0
    return bench, c


def test_timed_records_into_slots_and_reads_after_the_barrier():
    bench, c = _ctx(5, 2)
    log = []
    c.barrier = lambda: log.append("barrier")
    sec = bench.Ctx.timed(c, lambda: log.append("step"), after_warmup=lambda: log.append("aw"),
                          slot=lambda k: log.append(f"s{k}"), read_slot=lambda: log.append("read"))
    assert sec >= 0.0
    assert log == (["step", "step", "aw", "barrier"] + [x for k in range(5) for x in (f"s{k}", "step")] + ["barrier"] +
                   [x for k in range(5) for x in (f"s{k}", "read")] + ["s0"])


def test_timed_more_steps_than_slots_reads_the_last_launches():
    bench, c = _ctx(bench_steps := 70, 0)
    log = []
    c.barrier = lambda: None
    bench.Ctx.timed(c, lambda: None, slot=lambda k: log.append(k), read_slot=lambda: log.append("r"))
    rec, rd = log[:bench_steps], log[bench_steps:]
    assert rec == [k % bench.PROFILE_SLOTS for k in range(bench_steps)]
    assert rd.count("r") == bench.PROFILE_SLOTS and rd[-1] == 0


def test_timed_without_slots_is_the_plain_contract():
    bench, c = _ctx(3, 1)
    log = []
    c.barrier = lambda: log.append("barrier")
    bench.Ctx.timed(c, lambda: log.append("step"), per_step=lambda: log.append("ps"))
    assert log == ["step", "barrier", "step", "ps", "step", "ps", "step", "ps", "barrier"]
