#!/bin/bash
# round 4, run 15: where the 240 ms of the Zipf-like column go (run 14) -- rocprofv3 kernel stats of that line and of the normal keys
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out; R=15; L=$O/r4_run15.log; : > $L
prof() { local tag=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$tag" -o $tag -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 2 --warmup 1 "$@") > $O/r4_run${R}_bench_${tag}_under_rocprof.jsonl 2>> $L
  db=$(find $O/prof_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 4 run $R: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 $*" | head -40 | cut -c1-190 > $O/r4_run${R}_${tag}_kernel_stats.txt
  find $O/prof_$tag -name "*.db" -delete
}
prof sort_zipf --workload sort --key-dist zipf
prof sort_normal --workload sort --key-dist normal
head -30 $O/r4_run${R}_sort_zipf_kernel_stats.txt | cut -c1-175
head -22 $O/r4_run${R}_sort_normal_kernel_stats.txt | cut -c1-175
tail -3 $L
