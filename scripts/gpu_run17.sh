#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run17.log
: > $L
for w in reduce scan gather; do
  python bench.py --workload $w --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $L 2>&1
done
python bench.py --workload sorted_order --no-cpu-baseline >> $L 2>&1
grep -h '"metric"' $L > $O/bench17.jsonl
python -c "
import sys, json
for l in open('$O/bench17.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:80], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '| GB/s', round(r.get('achieved',0)), '| frac', round(r.get('frac',0),3))
"
tail -3 $L | cut -c1-300
