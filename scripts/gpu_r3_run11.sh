#!/bin/bash
# round 3 run 11: kernel trace of the default sort bench (cursor path): what do the skipped kernels and memsets cost?
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run11.log
: > $L
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_sort" -o sort -- python "$GRAFT_REPO_ROOT/bench.py" --workload sort --steps 5 --warmup 2 --no-cpu) 2>&1 | grep -v "simple_timer\|generateRocpd\|tool.cpp" | tail -3 >> $L
db=$(find $O/prof_sort -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/rocprof_summary.py "$db" "round 3 run 11: rocprofv3 --kernel-trace --stats -- python bench.py --workload sort --steps 5 --warmup 2 --no-cpu (cursor path)" | head -40 | cut -c1-220 > $O/r3_run11_sort_kernel_stats.txt
  cat $O/r3_run11_sort_kernel_stats.txt | tee -a $L
fi
find $O/prof_sort -name "*.db" -delete
echo finished | tee -a $L
