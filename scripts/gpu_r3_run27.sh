#!/bin/bash
# round 3 run 27: k_local_sort / k_local_place as strided loops over the cells (k_local_sort: 4096 workgroups over the todo list;
# k_local_place: one workgroup per cell by default, persistent grids as an A/B) -- sort tests, then the A/B at 1e9 rows
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run27.log
: > $L
: > $O/r3_run27_place_grid_ab.txt
for g in 0 2048 8192 32768; do
  PLACE_GRID=$g timeout 120 python scripts/xp/xp_place_ab.py 1e9 keys i64 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run27_place_grid_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort.py tests/test_gpu_parity_1e8.py -m gpu -q -x -k "sort or order or place or cursor" > $O/r3_run27_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -6 $O/r3_run27_pytest.log | tee -a $L
