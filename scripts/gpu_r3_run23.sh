#!/bin/bash
# round 3 run 23: k_local_place for both cell sizes (float keys and pairs above 1.02e9 rows: 16384-key cells), idle waves of
# part-filled cells skip the networks -- parity, then A/B at size: int64 1e9 / 1.25e9, float64 1e9 (keys and sorted_order)
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run23.log
: > $L
timeout 600 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py "tests/test_gpu_sort.py" -m gpu -q -x -k "place or cursor or hybrid or placed or crowded or knob or through or capacity" > $O/r3_run23_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/r3_run23_pytest.log | tee -a $L
: > $O/r3_run23_place_ab.txt
timeout 200 python scripts/xp/xp_place_ab.py 1e9 keys i64 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run23_place_ab.txt
timeout 200 python scripts/xp/xp_place_ab.py 1.25e9 keys i64 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run23_place_ab.txt
timeout 300 python scripts/xp/xp_place_ab.py 1e9 both f64 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run23_place_ab.txt
