#!/bin/bash
# round 2 run 14: hash-and-verify multi-column keys (Python + C++), pooled mr test fix, partitioned MIN/MAX tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run14.log
: > $L
( time timeout 1500 python -m pytest tests/test_cpp_api.py tests/test_gpu_join_kinds_multikey.py tests/test_gpu_join_groupby.py tests/test_gpu_dataframe.py -m gpu -q ) > $O/pytest_gpu14.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real|CHECK failed|FAIL\]" $O/pytest_gpu14.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu14.log | head -30 | tee -a $L
for w in join_multikey groupby_multikey; do
  ( time timeout 900 python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 ) > $O/bench14_$w.jsonl 2>> $L
done
python - <<'PY'
import json
for f in ('join_multikey', 'groupby_multikey'):
    try:
        d = json.loads(open(f'gpurun_out/bench14_{f}.jsonl').read().strip().split('\n')[-1])
        print(f, round(d['ms_per_step'], 3), 'ms', round(d['roofline']['frac'], 3), d['config']['workload'])
    except Exception as e:
        print(f, 'failed', e)
PY
tail -12 $L
