#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run14.log
: > $L
for k in 16 12 8; do
  echo "== MSD_KPT $k" >> $L
  GX_MSD_KPT=$k python bench.py --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $L 2>&1
done
GX_MSD_KPT=8 timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -x -q -k hybrid 2>&1 | tail -2 >> $L
grep -h '"metric"\|^==\|passed\|failed' $L | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    d = json.loads(l); r = d['roofline']
    print('   ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '|', {k[:18]: round(v,2) for k,v in r.get('kernels_ms',{}).items()})
"
