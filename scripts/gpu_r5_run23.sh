#!/bin/bash
# round 5 run 23: X passes with one ticket per workgroup when X is small -- the big-cell / splitter / float / fault tests, then the evidence run
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_sort_fault.py tests/test_gpu_sort_big_cells.py tests/test_cpp_api.py -m gpu -q -x 2>&1 | tail -12 ) > $O/r5_run23_tests.log
tail -n 5 $O/r5_run23_tests.log
if grep -q "failed\|error" $O/r5_run23_tests.log; then echo "TESTS FAILED: no evidence run"; exit 1; fi
bash scripts/gpu_r5_evidence.sh 23 pmc 2>&1 | tail -25 | cut -c1-250
