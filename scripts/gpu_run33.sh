#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run33.log
: > $L
python bench.py --key-range 100 10001 --no-cpu-baseline > $O/bench33_lowentropy.jsonl 2>> $L
timeout 200 python -m pytest tests/test_gpu_join_kinds_multikey.py -m gpu -q -k "semi_anti" > $O/pytest_gpu33.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -3 $O/pytest_gpu33.log | tee -a $L
cat $O/bench33_lowentropy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:90], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '| frac', round(r.get('frac',0),3), r.get('sort_info'))
"
