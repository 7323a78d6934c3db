#!/bin/bash
# round 2 run 8: persistent prefetching partition passes (k_msd_pass): sort tests, 1e8 parity, bench, SQ counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run8.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_parity_1e8.py -m gpu -q -x -k "sort" > $O/pytest_gpu8.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu8.log | head -20 | tee -a $L
grep -E "^E  " $O/pytest_gpu8.log | head -20 | tee -a $L
: > $O/bench8_sort.jsonl
timeout 300 python bench.py --workload sort --no-cpu-baseline >> $O/bench8_sort.jsonl 2>> $L
timeout 300 python bench.py --workload sorted_order --no-cpu-baseline >> $O/bench8_sort.jsonl 2>> $L
timeout 300 python bench.py --workload sort --no-cpu-baseline --rows 2e8 >> $O/bench8_sort.jsonl 2>> $L
python - <<'PY'
import json
for l in open('gpurun_out/bench8_sort.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '|', round(d['ms_per_step'], 2), 'ms | hist', round(r.get('hist_kernel_ms', 0), 2), '|',
          [round(v, 2) for v in r.get('kernels_ms', {}).values()], r.get('sort_info'))
PY
grep -E "exit|Error|error" $L | head -20
