#!/bin/bash
# round 2 run 12: DPP wave scans everywhere + 16384-element look-back tiles: whole GPU suite, scan / reduce bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run12.log
: > $L
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu12.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real" $O/pytest_gpu12.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu12.log | head -30 | tee -a $L
( time timeout 600 python bench.py --workload scan --no-cpu-baseline ) > $O/bench12_scan.jsonl 2>> $L
( time timeout 600 python bench.py --workload reduce --no-cpu-baseline ) > $O/bench12_reduce.jsonl 2>> $L
python - <<'PY'
import json
for f in ('scan', 'reduce'):
    for line in open(f'gpurun_out/bench12_{f}.jsonl').read().strip().split('\n'):
        try:
            d = json.loads(line)
        except Exception:
            print(line[:200]); continue
        print(f, d['config']['workload'][:60], round(d['ms_per_step'], 3), 'ms', round(d['roofline']['frac'], 3))
PY
