#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run18.log
: > $L
timeout 900 python -m pytest tests/test_gpu_reduce_scan_hash.py tests/test_gpu_join_groupby.py tests/test_gpu_sort.py -m gpu -x -q > $O/pytest_gpu18.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -4 $O/pytest_gpu18.log | tee -a $L
for w in reduce scan; do
  python bench.py --workload $w --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $L 2>&1
done
grep -h '"metric"' $L > $O/bench18.jsonl
python -c "
import sys, json
for l in open('$O/bench18.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:80], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '| GB/s', round(r.get('achieved',0)), '| frac', round(r.get('frac',0),3))
"
