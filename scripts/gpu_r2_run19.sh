#!/bin/bash
# round 2 run 19: what the driver runs at round end -- the whole GPU suite, smoke, the default bench line -- plus --through-cpp
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run19.log
: > $L
( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest_gpu19.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real" $O/pytest_gpu19.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu19.log | head -30 | tee -a $L
python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1; echo "smoke exit $?" | tee -a $L
( time timeout 900 python bench.py ) > $O/bench19_default.jsonl 2>> $L
( time timeout 900 python bench.py --no-cpu-baseline --through-cpp ) > $O/bench19_through_cpp.jsonl 2>> $L
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench19_default.jsonl').read().strip().split('\n')[-1])
print('sort', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e9, 1), 'Grows/s; dominant', d['roofline']['kernel'][:30], round(d['roofline']['frac'], 3), 'path_frac', round(d['roofline']['path_frac'], 3), 'traffic', d['roofline']['traffic'])
for k in ('join', 'groupby'):
    b = d[k]; print(k, round(b['ms_per_step'], 2), 'ms', round(b['value'] / 1e9, 1), 'Grows/s frac', round(b['roofline']['frac'], 3), 'traffic', b['roofline'].get('traffic'))
t = json.loads(open('gpurun_out/bench19_through_cpp.jsonl').read().strip().split('\n')[-1])['through_cpp']
print('through_cpp', {k: round(v, 3) if isinstance(v, float) else v for k, v in t.items() if k != 'note'})
PY
grep -v amdgpu.ids $L | grep -E "real|exit" | head
