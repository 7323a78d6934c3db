#!/bin/bash
# round 3 run 29: 32-bit integer keys on the cursor path -- parity (uniform / duplicates / narrow / sorted / skewed / outlier), every
# sort test again (k_hf_scatter and k_local_place were touched), then int32 at 1e9 rows against the LSD passes
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run29.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort.py -m gpu -q -x > $O/r3_run29_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -12 $O/r3_run29_pytest.log | tee -a $L
timeout 200 python scripts/xp/xp_sort32.py 1e9 2>&1 | grep -v amdgpu.ids | tee $O/r3_run29_sort32.txt
timeout 100 python scripts/xp/xp_place_ab.py 1e9 keys i64 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/r3_run29_sort32.txt
