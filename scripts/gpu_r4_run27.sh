#!/bin/bash
# round 4, run 27: the look-back path's level 1 asks for its bucket's cell-slot entries early too -- every sort test, loopback sort,
# sort-using C++ parity tests, then the evidence run on the final kernels
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py -q -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "sort" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity_1e8.py tests/test_gpu_cpp_parity.py -x -q -k "sort or rank or top_k or segmented or scan" 2>&1 | tail -3
} > $O/r4_run27_tests.log 2>&1
cat $O/r4_run27_tests.log
bash scripts/gpu_r4_evidence.sh 27 pmc no robust
