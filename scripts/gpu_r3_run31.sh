#!/bin/bash
# round 3 run 31: k_local_place's second (interpolating) counting pass for crowded cells, dense (key, row) words for sorted_order
# of 32-bit keys -- every sort test, then sorted_order int32 at 1e9 rows (wide keys / 1e6 dense / 1e6 sparse), int64 A/B lines
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run31.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort.py -m gpu -q -x > $O/r3_run31_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -12 $O/r3_run31_pytest.log | tee -a $L
timeout 150 python scripts/xp/xp_order32.py 1e9 2>&1 | grep -v amdgpu.ids | tee $O/r3_run31_order32.txt
timeout 150 python scripts/xp/xp_order32.py 1e9 1e6 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run31_order32.txt
timeout 150 python scripts/xp/xp_order32.py 1e9 sparse1e6 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run31_order32.txt
timeout 150 python scripts/xp/xp_place_ab.py 1e9 both i64 2>&1 | grep -v amdgpu.ids | grep "exp= 0" | tail -2 | tee -a $O/r3_run31_order32.txt
