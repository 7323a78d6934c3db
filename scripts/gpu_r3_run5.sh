#!/bin/bash
# round 3 run 5: the sharded operators in C++ over RCCL (gxd) on a 1-rank communicator with the exchange forced: GPU tests
# (Python caller + the C++ test case), then the per-rank step costs at 1e9 rows that the scaling model of DESIGN.md uses
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run5.log
: > $L
timeout 900 python -m pytest tests/test_gpu_distributed.py -q --durations=5 > $O/r3_run5_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -25 $O/r3_run5_pytest.log | tee -a $L
timeout 300 tests/cpp/cudf_api_tests > $O/r3_run5_cpp.log 2>&1
echo "cpp tests exit $?" | tee -a $L
tail -6 $O/r3_run5_cpp.log | tee -a $L
timeout 900 python scripts/xp/xp_distributed_single_rank.py 1e9 > $O/r3_xp_distributed_single_rank.txt 2>> $L
cat $O/r3_xp_distributed_single_rank.txt | tee -a $L
