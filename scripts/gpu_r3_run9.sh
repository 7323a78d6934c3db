#!/bin/bash
# round 3 run 9: sharded join with encoded row codes (no gathers) -- tests first, then the forced single-rank steps + kernel trace
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run9.log
: > $L
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_join_partition_modes.py tests/test_abi_symbols.py -q -x > $O/r3_run9_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -8 $O/r3_run9_pytest.log | tee -a $L
(timeout 600 tests/cpp/cudf_api_tests 2>&1 | tail -6) | tee -a $L
timeout 600 python scripts/xp/xp_distributed_single_rank.py 1e9 2>&1 | tail -30 | tee $O/r3_run9_single_rank_steps.txt | tee -a $L
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_gxdjoin" -o gxdjoin -- python "$GRAFT_REPO_ROOT/scripts/xp/xp_gxd_join_trace.py" 1e9 4) 2>&1 | grep -v "simple_timer\|generateRocpd\|tool.cpp" >> $L
db=$(find $O/prof_gxdjoin -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/rocprof_summary.py "$db" "round 3 run 9: rocprofv3 --kernel-trace --stats -- xp_gxd_join_trace.py 1e9 4 (build + 3 probes), row codes instead of gathers" | head -30 | cut -c1-200 > $O/r3_run9_gxd_join_kernel_stats.txt
  cat $O/r3_run9_gxd_join_kernel_stats.txt | tee -a $L
fi
find $O/prof_gxdjoin -name "*.db" -delete
echo finished | tee -a $L
