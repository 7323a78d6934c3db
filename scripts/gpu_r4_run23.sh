#!/bin/bash
# round 4, run 23: level 1 reads the column's own region table at static addresses again (the external form only for the sharded
# sort) -- cursor-path, big-cell, placement and loopback-sort tests, then the evidence run on the final kernels
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py tests/test_gpu_sort_place.py -q -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "sort" 2>&1 | tail -3
} > $O/r4_run23_tests.log 2>&1
cat $O/r4_run23_tests.log
bash scripts/gpu_r4_evidence.sh 23 pmc no robust
