#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run20.log
: > $L
timeout 900 python -m pytest tests/test_gpu_join_groupby.py tests/test_cpp_api.py tests/test_gpu_dataframe.py -m gpu -x -q > $O/pytest_gpu20.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/pytest_gpu20.log | tee -a $L
