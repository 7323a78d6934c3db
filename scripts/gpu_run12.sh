#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run12.log
: > $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 12: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run12_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof join_nt --workload join --rows 1e9 --steps 2 --warmup 1
cat $O/r1_run12_*_kernel_stats.txt | grep -E "^# round|k_pj|k_probe|k_build" | cut -c1-170
