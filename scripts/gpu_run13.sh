#!/bin/bash
# full GPU suite + default bench (with cpu_baseline) + rocprof of the default command
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run22.log
: > $L
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu13.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -6 $O/pytest_gpu13.log | tee -a $L
python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1; echo "smoke exit $?" | tee -a $L
( time python bench.py ) > $O/bench13_default.jsonl 2>> $L
python bench.py --workload sorted_order --no-cpu-baseline >> $O/bench13_others.jsonl 2>> $L
python bench.py --workload join --no-cpu-baseline >> $O/bench13_others.jsonl 2>> $L
python bench.py --workload groupby --no-cpu-baseline >> $O/bench13_others.jsonl 2>> $L
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_default" -o default -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) >> $L 2>&1
db=$(find $O/prof_default -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 22: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  (default: 1e9-row int64 sort, 5 steps + 2 warmup)" > $O/r1_run22_default_kernel_stats.txt
find $O/prof_default -name "*.db" -delete
cat $O/bench13_default.jsonl | cut -c1-3000
cat $O/bench13_others.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '| frac', round(r.get('frac',0),3))
"
head -12 $O/r1_run22_default_kernel_stats.txt | cut -c1-170
grep -E "real|smoke" $L
