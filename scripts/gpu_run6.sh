#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run7.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort.py -m gpu -x -q -k "hybrid" > $O/pytest_gpu6.log 2>&1
echo "pytest hybrid exit $?" | tee -a $L
tail -30 $O/pytest_gpu6.log | tee -a $L
timeout 900 python -m pytest tests/test_gpu_sort.py -m gpu -x -q > $O/pytest_gpu6b.log 2>&1
echo "pytest sort exit $?" | tee -a $L
tail -5 $O/pytest_gpu6b.log | tee -a $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 7: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run7_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof sort_hybrid --rows 1e9 --steps 2 --warmup 1
prof sort_lsd --rows 1e9 --steps 2 --warmup 1 --no-hybrid
prof sorted_order --workload sorted_order --rows 1e9 --steps 2 --warmup 1
cat $O/r1_run7_*_kernel_stats.txt | grep -E "^# round|k_radix_pass|k_hist|k_msd|k_local|k_plan" | cut -c1-170
grep -h '"metric"' $L | cut -c1-1800
tail -3 $L
