#!/bin/bash
# round 3 run 25: 16-bit per-wave counters only where they buy the second workgroup (pairs, 9-bit level 1); the rest of the GPU
# suite (run 24 stopped at a stale assertion of the multi-key collision test), A/B of sorted_order
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run25.log
: > $L
timeout 200 python scripts/xp/xp_place_ab.py 1e9 order i64 2>&1 | grep -v amdgpu.ids | tee $O/r3_run25_place_ab.txt
timeout 1200 python -m pytest tests/test_gpu_join_kinds_multikey.py tests/test_gpu_join_partition_modes.py tests/test_gpu_parity_1e8.py tests/test_gpu_reduce_scan_hash.py tests/test_gpu_sort.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_place.py -m gpu -q --durations=5 > $O/r3_run25_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -14 $O/r3_run25_pytest.log | tee -a $L
