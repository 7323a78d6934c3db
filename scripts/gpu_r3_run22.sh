#!/bin/bash
# round 3 run 22: k_local_place (counting placement + per-thread window networks) -- parity of every hybrid-path sort test,
# then the A/B at 1e9 rows against k_local_sort's sub-bucket path (knob 32), keys only and sorted_order
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run22.log
: > $L
timeout 600 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py "tests/test_gpu_sort.py" -m gpu -q -x -k "place or cursor or hybrid or placed or crowded or knob or sorted_order_through" > $O/r3_run22_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/r3_run22_pytest.log | tee -a $L
timeout 300 python scripts/xp/xp_place_ab.py 1e9 both 2>&1 | grep -v amdgpu.ids | tee $O/r3_run22_place_ab.txt
