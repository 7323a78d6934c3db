#!/bin/bash
# round 2 run 25: LDS groupby kernels with the probe / spill slow paths out of line: tests + groupby benches
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run25.log
: > $L; : > $O/bench25_groupby.jsonl
( time timeout 1500 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_join_kinds_multikey.py tests/test_gpu_dataframe.py tests/test_gpu_parity_1e8.py tests/test_cpp_api.py -m gpu -q -k "groupby or cpp_api or dataframe" ) > $O/pytest_gpu25.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu25.log | head | tee -a $L
grep -E "^E  " $O/pytest_gpu25.log | head -20 | tee -a $L
for w in groupby groupby_minmax groupby_multikey; do
  ( timeout 900 python bench.py --workload $w --no-cpu-baseline ) >> $O/bench25_groupby.jsonl 2>> $L
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_gb25" -o gb -- python "$GRAFT_REPO_ROOT/bench.py" --workload groupby --no-cpu-baseline --steps 3 --warmup 1) >> $L 2>&1
db=$(find $O/prof_gb25 -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 2 run 25: groupby" | grep "gx::gb" | head -5 | cut -c1-200
find $O/prof_gb25 -name "*.db" -delete
python - <<'PY'
import json
for line in open('gpurun_out/bench25_groupby.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    print(d['config']['workload'][:70], round(d['ms_per_step'], 2), 'ms')
PY
