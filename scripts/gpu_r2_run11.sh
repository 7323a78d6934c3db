#!/bin/bash
# round 2 run 11: look-back scan validation + scan bench, partition_rows fix
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run11.log
: > $L
( time timeout 1200 python -m pytest tests/test_gpu_reduce_scan_hash.py tests/test_gpu_distributed.py tests/test_gpu_dataframe.py -m gpu -q -x ) > $O/pytest_gpu11.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real" $O/pytest_gpu11.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu11.log | head -30 | tee -a $L
( time timeout 600 python bench.py --workload scan --no-cpu-baseline ) > $O/bench11_scan.jsonl 2>> $L
( time timeout 600 python bench.py --workload reduce --no-cpu-baseline ) > $O/bench11_reduce.jsonl 2>> $L
python - <<'PY'
import json
for f in ('scan', 'reduce'):
    for line in open(f'gpurun_out/bench11_{f}.jsonl').read().strip().split('\n'):
        try:
            d = json.loads(line)
        except Exception:
            print(line[:200]); continue
        print(f, d['config']['workload'][:60], round(d['ms_per_step'], 3), 'ms', d['roofline'])
PY
