#!/bin/bash
# round 2 run 10: the whole GPU suite as the driver runs it, smoke, default bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run10.log
: > $L
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > $O/pytest_gpu10.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real" $O/pytest_gpu10.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu10.log | head -30 | tee -a $L
grep -A14 "slowest" $O/pytest_gpu10.log | tee -a $L
python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1; echo "smoke exit $?" | tee -a $L
( time timeout 900 python bench.py ) > $O/bench10_default.jsonl 2>> $L
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench10_default.jsonl').read().strip().split('\n')[-1])
print('sort', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e9, 1), 'Grows/s; dominant', d['roofline']['kernel'][:30], round(d['roofline']['frac'], 3), 'path_frac', round(d['roofline']['path_frac'], 3))
for k in ('join', 'groupby'):
    b = d[k]; print(k, round(b['ms_per_step'], 2), 'ms', round(b['value'] / 1e9, 1), 'Grows/s frac', round(b['roofline']['frac'], 3), b['roofline'].get('kernels_ms'))
PY
grep -E "real|exit|Error|error" $L | head -20
