#!/bin/bash
# round 3 run 32: one more level-1 bit on the device for fuller buckets (key ranges that are not a power of two), the word path of
# sorted_order(int32) behind a knob, second counting pass of k_local_place -- every sort test, then timings at 1e9 rows
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run32.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort.py -m gpu -q -x > $O/r3_run32_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -12 $O/r3_run32_pytest.log | tee -a $L
timeout 200 python scripts/xp/xp_sort_range.py 1e9 2>&1 | grep -v amdgpu.ids | grep "cursor_path=1" | tee $O/r3_run32_sort_range.txt
timeout 150 python scripts/xp/xp_place_ab.py 1e9 keys i64 2>&1 | grep -v amdgpu.ids | grep "exp= 0" | tail -2 | tee -a $O/r3_run32_sort_range.txt
timeout 100 python scripts/xp/xp_sort32.py 1e9 2>&1 | grep -v amdgpu.ids | grep "cursor_path=1" | tail -1 | tee -a $O/r3_run32_sort_range.txt
