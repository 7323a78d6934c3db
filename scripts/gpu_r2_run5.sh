#!/bin/bash
# round 2 run 5: two-way interleaved sub-bucket sort + DPP scans + parallel cell plan; sort-path groupby, rank/top_k/segmented
# sort, shift / get_groups / replace_nulls through the C++ API
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run5.log
: > $L
timeout 1200 python -m pytest tests/test_cpp_api.py tests/test_gpu_sort.py -m gpu -q > $O/pytest_gpu5.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|CHECK failed|FAIL\]|what\(\)" $O/pytest_gpu5.log | head -40 | tee -a $L
: > $O/bench5_sort.jsonl
timeout 300 python bench.py --workload sort --no-cpu-baseline >> $O/bench5_sort.jsonl 2>> $L
timeout 300 python bench.py --workload sort --no-cpu-baseline --sort-cell 16384 >> $O/bench5_sort.jsonl 2>> $L
timeout 300 python bench.py --workload sorted_order --no-cpu-baseline >> $O/bench5_sort.jsonl 2>> $L
python - <<'PY'
import json
for l in open('gpurun_out/bench5_sort.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '|', round(d['ms_per_step'], 2), 'ms | hist', round(r.get('hist_kernel_ms', 0), 2), '|',
          [round(v, 2) for v in r.get('kernels_ms', {}).values()], r.get('sort_info'))
PY
grep -E "exit|Error|error" $L | head -20
