#!/bin/bash
# round 2 run 2: new hybrid sort plan (bit-granular digits, single-digit up-front histogram, 9-bit level 1 +
# 8192-key cells): sort tests, then A/B of the cell size, narrow-range / shard-like keys, sorted_order
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run2.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_sort.py -m gpu -x -q > $O/pytest_gpu2.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/pytest_gpu2.log | tee -a $L
: > $O/bench2_sort_ab.jsonl
for cell in 0 16384; do
  timeout 300 python bench.py --workload sort --no-cpu-baseline --sort-cell $cell >> $O/bench2_sort_ab.jsonl 2>> $L
  echo "sort cell=$cell exit $?" >> $L
done
timeout 300 python bench.py --workload sort --no-cpu-baseline --key-range 0 1152921504606846976 >> $O/bench2_sort_ab.jsonl 2>> $L   # 2^60: top 4 bits constant (a rank's shard)
timeout 300 python bench.py --workload sort --no-cpu-baseline --rows 1.05e9 >> $O/bench2_sort_ab.jsonl 2>> $L
timeout 300 python bench.py --workload sorted_order --no-cpu-baseline >> $O/bench2_sort_ab.jsonl 2>> $L
timeout 300 python bench.py --workload sort --no-cpu-baseline --key-range 100 10001 >> $O/bench2_sort_ab.jsonl 2>> $L
python - <<'PY'
import json
for l in open('gpurun_out/bench2_sort_ab.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:70], '|', round(d['ms_per_step'], 2), 'ms | hist', round(r.get('hist_kernel_ms', 0), 2), '|',
          [round(v, 2) for v in r.get('kernels_ms', {}).values()], r.get('sort_info'))
PY
grep -E "exit|Error|error" $L | head -20
