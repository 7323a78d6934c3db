#!/bin/bash
# round 2 run 17: row keys (packed / certified hash) in ops, DataFrame and C++; nsub LDS tables; headline unchanged?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run17.log
: > $L
( time timeout 1500 python -m pytest tests/test_cpp_api.py tests/test_gpu_join_kinds_multikey.py tests/test_gpu_join_groupby.py tests/test_gpu_dataframe.py tests/test_gpu_parity_1e8.py -m gpu -q ) > $O/pytest_gpu17.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real|CHECK failed|FAIL\]" $O/pytest_gpu17.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu17.log | head -30 | tee -a $L
for w in groupby groupby_minmax join_multikey groupby_multikey; do
  ( timeout 900 python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 ) > $O/bench17_$w.jsonl 2>> $L
done
timeout 600 python scripts/xp/xp_minmax_matrix.py 2.5e8 2>&1 | grep -v amdgpu.ids | grep "nsplit 1 " > $O/xp_minmax_matrix_nsub.txt
python - <<'PY'
import json
for f in ('groupby', 'groupby_minmax', 'join_multikey', 'groupby_multikey'):
    try:
        d = json.loads(open(f'gpurun_out/bench17_{f}.jsonl').read().strip().split('\n')[-1])
        print(f, round(d['ms_per_step'], 3), 'ms', round(d['roofline']['frac'], 3), d['config']['workload'])
    except Exception as e:
        print(f, 'failed', e)
PY
cat $O/xp_minmax_matrix_nsub.txt
grep -v amdgpu.ids $L | tail -8
