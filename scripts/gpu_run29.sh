#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run29.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_join_kinds_multikey.py tests/test_gpu_dataframe.py tests/test_cpp_api.py tests/test_gpu_join_groupby.py -m gpu -q > $O/pytest_gpu29.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu29.log | head -40 | tee -a $L
grep -E "FAIL\]|CHECK failed" $O/pytest_gpu29.log | head -20 | tee -a $L
