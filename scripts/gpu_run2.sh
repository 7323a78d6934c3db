#!/bin/bash
# round 1, GPU call 2: parity suite + A/B of the look-back window, partitioned groupby, full-size join
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run2.log
: > $L
echo "== pytest -m gpu" | tee -a $L
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu2.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -8 $O/pytest_gpu2.log | tee -a $L
B=$O/bench2.jsonl
: > $B
for algo in 0 2 1; do
  timeout 300 python bench.py --rows 1e9 --steps 3 --warmup 1 --algo $algo --no-cpu-baseline >> $B 2>> $L
done
timeout 300 python bench.py --workload sorted_order --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $B 2>> $L
timeout 300 python bench.py --workload groupby --rows 1e9 --steps 3 --warmup 1 --gb-algo 0 --no-cpu-baseline >> $B 2>> $L
timeout 300 python bench.py --workload groupby --rows 1e9 --steps 3 --warmup 1 --gb-algo 0 --gb-split 4 --no-cpu-baseline >> $B 2>> $L
timeout 300 python bench.py --workload groupby --rows 1e9 --steps 2 --warmup 1 --gb-algo 1 --no-cpu-baseline >> $B 2>> $L
timeout 300 python bench.py --workload join --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $B 2>> $L
echo "== rocprof" | tee -a $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 2: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run2_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof sort_a0 --rows 1e9 --steps 2 --warmup 1 --algo 0
prof sort_a1 --rows 1e9 --steps 2 --warmup 1 --algo 1
prof groupby --workload groupby --rows 1e9 --steps 2 --warmup 1
prof join --workload join --rows 1e9 --steps 2 --warmup 1
cat $B
tail -30 $L
