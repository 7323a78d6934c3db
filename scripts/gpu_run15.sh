#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run15.log
: > $L
timeout 900 python -m pytest tests/test_gpu_sort.py -m gpu -x -q > $O/pytest_gpu15.log 2>&1
echo "pytest sort exit $?" | tee -a $L
tail -12 $O/pytest_gpu15.log | tee -a $L
python bench.py --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $L 2>&1
python bench.py --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline --workload sorted_order >> $L 2>&1
grep -h '"metric"' $L | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:50], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '|', r['kernel'][:30], round(r['avg_launch_ms'],2), '| hist', round(r.get('hist_kernel_ms',0),2), '|', {k[:18]: round(v,2) for k,v in r.get('kernels_ms',{}).items()}, r.get('sort_info'))
"
