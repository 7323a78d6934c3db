#!/bin/bash
# round 2 run 9: msd pass with early (asynchronous) ticket, atomic optimizer off: sort tests + benches (sort, join, groupby)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run9.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_join_groupby.py -m gpu -q -x > $O/pytest_gpu9.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu9.log | head -20 | tee -a $L
grep -E "^E  " $O/pytest_gpu9.log | head -20 | tee -a $L
: > $O/bench9.jsonl
timeout 300 python bench.py --workload sort --no-cpu-baseline >> $O/bench9.jsonl 2>> $L
timeout 300 python bench.py --workload join --no-cpu-baseline >> $O/bench9.jsonl 2>> $L
timeout 300 python bench.py --workload groupby --no-cpu-baseline >> $O/bench9.jsonl 2>> $L
python - <<'PY'
import json
for l in open('gpurun_out/bench9.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '|', round(d['ms_per_step'], 2), 'ms | hist', round(r.get('hist_kernel_ms', 0), 2), '|',
          [round(v, 2) for v in r.get('kernels_ms', {}).values()])
PY
grep -E "exit|Error|error" $L | head -20
