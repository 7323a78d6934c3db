#!/bin/bash
# round 2 run 24: rocprofv3 --kernel-trace --stats of the non-default workloads (scan, reduce, groupby_minmax, sorted_order,
# gather, multi-column keys) -- the per-kernel times behind the DESIGN.md tables that the default command does not cover
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run24.log
: > $L; : > $O/r2_run24_other_workloads_kernel_stats.txt; : > $O/bench24_other_workloads.jsonl
for wl in scan reduce groupby_minmax sorted_order gather join_multikey groupby_multikey; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$wl" -o $wl -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --no-cpu-baseline --steps 3 --warmup 1) >> $O/bench24_other_workloads.jsonl 2>> $L
  db=$(find $O/prof_$wl -name "*.db" | head -1)
  if [ -n "$db" ]; then
    python scripts/rocprof_summary.py "$db" "round 2 run 24: rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1" | grep -E "^#|^kernel|gx::" | head -16 | cut -c1-190 >> $O/r2_run24_other_workloads_kernel_stats.txt
    echo >> $O/r2_run24_other_workloads_kernel_stats.txt
  fi
  find $O/prof_$wl -name "*.db" -delete
done
grep -c "gx::" $O/r2_run24_other_workloads_kernel_stats.txt
python - <<'PY'
import json
for line in open('gpurun_out/bench24_other_workloads.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    print(d['config']['workload'][:70], round(d['ms_per_step'], 2), 'ms')
PY
