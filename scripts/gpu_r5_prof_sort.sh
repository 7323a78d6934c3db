#!/bin/bash
# rocprofv3 kernel summary of one sort line: usage gpu_r5_prof_sort.sh <tag> <bench args...>
set -u
TAG=$1; shift
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$TAG" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --workload sort --no-cpu-baseline --steps 3 --warmup 1 "$@") > $O/r5_prof_${TAG}.jsonl 2> $O/r5_prof_${TAG}.log
db=$(find $O/prof_$TAG -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --workload sort --no-cpu-baseline --steps 3 --warmup 1 $*" | head -45 | cut -c1-175 > $O/r5_prof_${TAG}_kernel_stats.txt
find $O/prof_$TAG -name "*.db" -delete
cat $O/r5_prof_${TAG}_kernel_stats.txt
