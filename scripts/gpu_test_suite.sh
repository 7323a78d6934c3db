#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/runNN.log
: > $L
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpuNN.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpuNN.log | head -30 | tee -a $L
grep -A18 "slowest" $O/pytest_gpuNN.log | tee -a $L
