#!/bin/bash
# round 3 run 13: wave-aggregated ranks in the few-bin exchange partition; forced single-rank steps with the cursor-path local sort
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run13.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_join_partition_modes.py tests/test_gpu_join_groupby.py -q -x > $O/r3_run13_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/r3_run13_pytest.log | tee -a $L
(timeout 600 tests/cpp/cudf_api_tests 2>&1 | tail -3) | tee -a $L
GXD_TRACE=1 timeout 600 python scripts/xp/xp_distributed_single_rank.py 1e9 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -60 | tee $O/r3_run13_single_rank_steps.txt | tee -a $L
echo finished | tee -a $L
