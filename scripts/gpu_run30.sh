#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run30.log
: > $L
timeout 900 python -m pytest tests/test_cpp_api.py tests/test_gpu_dataframe.py tests/test_gpu_join_kinds_multikey.py -m gpu -q -k "cpp or groupby or compound or dense_rank" > $O/pytest_gpu30.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu30.log | head -40 | tee -a $L
grep -E "FAIL\]|CHECK failed|Error|error" $O/pytest_gpu30.log | head -30 | tee -a $L
