#!/bin/bash
# round 4, run 4: kernel traces -- the sort with 1e6 / 1e8 copies of one value (where the extra time goes), the join (build breakdown)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_sort_big_cells.py -x -q 2>&1 | tail -3 > $O/r4_run4_tests.log
prof() { # tag, bench args...
  local tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$tag" -o $tag -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 "$@") > $O/r4_run4_${tag}.jsonl 2> $O/r4_run4_${tag}.err
  db=$(find $O/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 4 run 4: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 $*" | head -40 | cut -c1-200 > $O/r4_run4_${tag}_kernel_stats.txt
  rm -rf $O/prof_$tag
}
prof sort_hot1e6 --workload sort --hot-copies 1e6
prof sort_hot1e8 --workload sort --hot-copies 1e8
prof join --workload join
cat $O/r4_run4_tests.log
for t in sort_hot1e6 sort_hot1e8 join; do echo "== $t"; grep -o '"ms_per_step": [0-9.]*' $O/r4_run4_${t}.jsonl | head -2; head -22 $O/r4_run4_${t}_kernel_stats.txt | cut -c1-60,105-170; done
