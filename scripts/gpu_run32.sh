#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run32.log
: > $L
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu32.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu32.log | head -30 | tee -a $L
grep -A18 "slowest" $O/pytest_gpu32.log | tee -a $L
