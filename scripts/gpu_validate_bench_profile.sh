#!/bin/bash
# validation of the last feature batch + smoke + benches + rocprof of the default command
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/runNN.log
: > $L
timeout 900 python -m pytest tests/test_cpp_api.py tests/test_gpu_dataframe.py tests/test_gpu_join_kinds_multikey.py -m gpu -q -k "cpp or groupby or compound" > $O/pytest_gpuNN.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpuNN.log | head -30 | tee -a $L
grep -E "FAIL\]|CHECK failed" $O/pytest_gpuNN.log | head -20 | tee -a $L
python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1; echo "smoke exit $?" | tee -a $L
( time python bench.py ) > $O/benchNN_default.jsonl 2>> $L
python bench.py --workload sorted_order --no-cpu-baseline >> $O/benchNN_others.jsonl 2>> $L
python bench.py --workload join --no-cpu-baseline >> $O/benchNN_others.jsonl 2>> $L
python bench.py --workload groupby --no-cpu-baseline >> $O/benchNN_others.jsonl 2>> $L
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_default" -o default -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline) >> $L 2>&1
db=$(find $O/prof_default -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round R run NN: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline  (default: 1e9-row int64 sort, 5 steps + 2 warmup)" > $O/r1_runNN_default_kernel_stats.txt
find $O/prof_default -name "*.db" -delete
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_join" -o join -- python "$GRAFT_REPO_ROOT/bench.py" --workload join --no-cpu-baseline) >> $L 2>&1
db=$(find $O/prof_join -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round R run NN: rocprofv3 --kernel-trace --stats -- python bench.py --workload join --no-cpu-baseline  (1e9 probe x 1e8 build)" > $O/r1_runNN_join_kernel_stats.txt
find $O/prof_join -name "*.db" -delete
cat $O/benchNN_default.jsonl | cut -c1-3000
cat $O/benchNN_others.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '| frac', round(r.get('frac',0),3), d.get('join_build_ms'))
"
head -12 $O/r1_runNN_default_kernel_stats.txt | cut -c1-170
head -10 $O/r1_runNN_join_kernel_stats.txt | cut -c1-170
grep -E "real|smoke" $L
