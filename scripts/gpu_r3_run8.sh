#!/bin/bash
# round 3 run 8: kernel trace of a forced gxd_join_probe step (where do the 37 ms go?) + the tests touched since run 7
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run8.log
: > $L
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_gxdjoin" -o gxdjoin -- python "$GRAFT_REPO_ROOT/scripts/xp/xp_gxd_join_trace.py" 1e9 4) 2>&1 | grep -v "simple_timer\|generateRocpd\|tool.cpp" >> $L
db=$(find $O/prof_gxdjoin -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/rocprof_summary.py "$db" "round 3 run 8: rocprofv3 --kernel-trace --stats -- xp_gxd_join_trace.py 1e9 4 (build + 3 probes)" | head -30 | cut -c1-200 > $O/r3_run8_gxd_join_kernel_stats.txt
  cat $O/r3_run8_gxd_join_kernel_stats.txt | tee -a $L
fi
find $O/prof_gxdjoin -name "*.db" -delete
timeout 600 python -m pytest tests/test_gpu_distributed.py tests/test_abi_symbols.py -q > $O/r3_run8_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/r3_run8_pytest.log | tee -a $L
