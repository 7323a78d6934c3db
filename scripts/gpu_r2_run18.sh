#!/bin/bash
# round 2 run 18: the evidence behind bench.py's numbers: rocprofv3 --kernel-trace --stats of the default command,
# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of sort / join / groupby / scan / reduce at 1e9 rows
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run18.log
: > $L
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_default" -o default -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) > $O/bench18_default_under_rocprof.jsonl 2>> $L
db=$(find $O/prof_default -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 2 run 18: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (sort + join + groupby, 5 steps + 2 warm-up each)" > $O/r2_run18_default_kernel_stats.txt
find $O/prof_default -name "*.db" -delete
pmc() { # workload, counter
  local wl=$1; local ctr=$2
  local lc=$(echo $ctr | tr 'A-Z' 'a-z')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_${wl}_${lc}" -o $wl --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline) >> $L 2>&1
}
for wl in sort join groupby scan reduce groupby_minmax; do
  pmc $wl FETCH_SIZE
  pmc $wl WRITE_SIZE
done
python scripts/pmc_to_json.py $O $O/r2_pmc_traffic_1e9.json | tee $O/r2_run18_pmc_traffic.txt
find $O/pmc_* -name "*.csv" -size +1M -delete
head -30 $O/r2_run18_default_kernel_stats.txt | cut -c1-200
tail -3 $O/bench18_default_under_rocprof.jsonl | cut -c1-400
grep -v amdgpu.ids $L | tail -5
