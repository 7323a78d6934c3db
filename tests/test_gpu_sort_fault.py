"""GPU: a broken look-back chain is REPORTED, not fatal.  Until round 5 a look-back wait that made no
progress for 30 s ended in __builtin_trap(): a queue exception the process does not survive.  Now the wait is abandoned, the sort's
status word (gx_sort_status) becomes 5, every write stays inside the output, and the host side raises -- the reference's analogue is
a recoverable error status out of cub (cpp/include/cudf/utilities/error.hpp:63-86 separates those from fatal ones).  The test hook
gx_sort_inject_lost_tile makes one tile of every look-back pass withhold its granules; gx_sort_set_spin_limit_ms cuts the wait."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hooks():
    from cudf_amd import _lib as L
    lib = L.lib
    lib.gx_sort_set_order_map(0)   # the look-back pairs levels are what this file breaks: above 2^25 rows sorted_order is a keys-only word sort by default (gx_order.hip)
    yield lib
    lib.gx_sort_inject_lost_tile(-1)
    lib.gx_sort_set_spin_limit_ms(0)
    lib.gx_sort_set_order_map(1)


@pytest.mark.parametrize("n,what", [(3_000_000, "lsd passes, keys only"), (3_000_000, "lsd passes, pairs"), (40_000_000, "hybrid levels, pairs")])
def test_lost_tile_is_reported_and_the_process_survives(hooks, n, what):
    import torch
    from cudf_amd import _lib as L, ops
    from oracle import cudf_oracle as oracle
    lib = hooks
    rng = np.random.default_rng(17)
    keys_h = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    keys = ops.Column.from_numpy(keys_h)
    lib.gx_sort_set_spin_limit_ms(150)
    lib.gx_sort_inject_lost_tile(3)
    with pytest.raises(L.GxError, match="look-back"):
        if "pairs" in what:
            ops.sorted_order(keys)
        else:
            ops.sort(keys)
    lib.gx_sort_inject_lost_tile(-1)
    lib.gx_sort_set_spin_limit_ms(0)
    torch.cuda.synchronize()
    # the same call right after: same process, same HIP context, right answer
    if "pairs" in what:
        got = ops.sorted_order(keys).to_numpy()
        np.testing.assert_array_equal(got, oracle.sorted_order(keys_h).astype(got.dtype))
    else:
        got = ops.sort(keys).to_numpy()
        np.testing.assert_array_equal(got, np.sort(keys_h))


def test_guard_does_not_fire_without_a_fault(hooks):
    """the shortened limit alone (no lost tile) changes nothing: slow predecessors are not faults, the limit is wall-clock time without progress"""
    from cudf_amd import ops
    lib = hooks
    rng = np.random.default_rng(18)
    keys_h = rng.integers(-2**62, 2**62, 6_000_000, dtype=np.int64)
    lib.gx_sort_set_spin_limit_ms(500)
    got = ops.sort(ops.Column.from_numpy(keys_h)).to_numpy()
    np.testing.assert_array_equal(got, np.sort(keys_h))
