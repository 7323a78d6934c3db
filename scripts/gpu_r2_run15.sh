#!/bin/bash
# round 2 run 15: per-kernel times of the multi-column-key paths (rocprofv3 --kernel-trace --stats), dataframe tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run15.log
: > $L
( time timeout 600 python -m pytest tests/test_gpu_dataframe.py -m gpu -q ) > $O/pytest_gpu15.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu15.log | head | tee -a $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 2 run 15: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r2_run15_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof groupby_multikey --workload groupby_multikey --steps 1 --warmup 0
prof join_multikey --workload join_multikey --steps 1 --warmup 0
head -40 $O/r2_run15_groupby_multikey_kernel_stats.txt
head -30 $O/r2_run15_join_multikey_kernel_stats.txt
