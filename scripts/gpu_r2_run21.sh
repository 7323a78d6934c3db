#!/bin/bash
# round 2 run 21: predecessors per look-back round in the hybrid partition passes: 4 (default) vs 8 vs 16
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run21.log
: > $L; : > $O/bench21_sort_lbw.jsonl
for lbw in 4 8 16 4 8 16; do
  ( timeout 600 python bench.py --workload sort --no-cpu-baseline --sort-lbw $lbw ) >> $O/bench21_sort_lbw.jsonl 2>> $L
done
python - <<'PY'
import json
for line in open('gpurun_out/bench21_sort_lbw.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    r = d['roofline']
    print(round(d['ms_per_step'], 2), 'ms', 'hist', round(r.get('hist_kernel_ms', 0), 2), [round(v, 2) for v in (r.get('kernels_ms') or {}).values()])
PY
grep -v amdgpu.ids $L | tail -3
