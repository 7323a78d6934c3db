#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run23.log
: > $L
timeout 900 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_dataframe.py tests/test_gpu_distributed.py -m gpu -x -q > $O/pytest_gpu23.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/pytest_gpu23.log | tee -a $L
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_join23" -o join23 -- python "$GRAFT_REPO_ROOT/bench.py" --workload join --rows 1e9 --steps 2 --warmup 1 --no-cpu-baseline) >> $L 2>&1
db=$(find $O/prof_join23 -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 23: rocprofv3 --kernel-trace --stats -- python bench.py --workload join --rows 1e9 --steps 2 --warmup 1" > $O/r1_run23_join_kernel_stats.txt
find $O/prof_join23 -name "*.db" -delete
grep -E "^# round|k_pj|k_probe|k_build" $O/r1_run23_join_kernel_stats.txt | cut -c1-170
grep -h '"metric"' $L | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('join ms', round(d['ms_per_step'],2), 'Grows/s', round(d['value']/1e9,2), 'build_ms', round(d.get('join_build_ms',0),2))
"
