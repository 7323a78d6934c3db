#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run5.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort.py tests/test_gpu_reduce_scan_hash.py -m gpu -x -q > $O/pytest_gpu4.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/pytest_gpu4.log | tee -a $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 5: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run5_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof sort_a0 --rows 1e9 --steps 2 --warmup 1 --algo 0
prof sort_a2 --rows 1e9 --steps 2 --warmup 1 --algo 2
prof sorted_order_a0 --workload sorted_order --rows 1e9 --steps 2 --warmup 1 --algo 0
cat $O/r1_run5_*_kernel_stats.txt | grep -E "^# round|k_radix_pass|k_hist" | cut -c1-170
grep -h '"metric"' $L | cut -c1-400
tail -3 $L
