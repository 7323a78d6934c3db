#!/bin/bash
# round 2 run 22: look-back window 16 by default, 16-byte loads in the groupby partition histogram: tests + default line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run22.log
: > $L
( time timeout 1500 python -m pytest tests/test_gpu_sort.py tests/test_gpu_parity_1e8.py tests/test_gpu_join_groupby.py tests/test_gpu_dataframe.py tests/test_cpp_api.py -m gpu -q ) > $O/pytest_gpu22.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu22.log | head | tee -a $L
grep -E "^E  " $O/pytest_gpu22.log | head -20 | tee -a $L
( timeout 900 python bench.py --no-cpu-baseline ) > $O/bench22_default.jsonl 2>> $L
( timeout 900 python bench.py --workload groupby --no-cpu-baseline ) >> $O/bench22_default.jsonl 2>> $L
( timeout 900 python bench.py --workload groupby_minmax --no-cpu-baseline ) >> $O/bench22_default.jsonl 2>> $L
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_gb" -o gb -- python "$GRAFT_REPO_ROOT/bench.py" --workload groupby --no-cpu-baseline --steps 3 --warmup 1) >> $L 2>&1
db=$(find $O/prof_gb -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 2 run 22: groupby" | grep "gx::" | head -8 | cut -c1-200
find $O/prof_gb -name "*.db" -delete
python - <<'PY'
import json
for line in open('gpurun_out/bench22_default.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    r = d['roofline']
    print(d['config']['workload'][:60], round(d['ms_per_step'], 2), 'ms', [round(v, 2) for v in (r.get('kernels_ms') or {}).values()])
    for k in ('join', 'groupby'):
        if k in d: print('   ', k, round(d[k]['ms_per_step'], 2))
PY
