#!/bin/bash
# round 2 run 13: pooled mr + async C++ surface, LDS-partitioned groupby MIN/MAX, through-C++ bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run13.log
: > $L
( time timeout 900 python -m pytest tests/test_cpp_api.py tests/test_gpu_join_groupby.py -m gpu -q -x -k "cpp_api or groupby" ) > $O/pytest_gpu13.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real|CHECK failed|FAIL\]" $O/pytest_gpu13.log | head -30 | tee -a $L
grep -E "^E  " $O/pytest_gpu13.log | head -20 | tee -a $L
( time timeout 600 python bench.py --workload groupby_minmax --no-cpu-baseline ) > $O/bench13_minmax.jsonl 2>> $L
( time timeout 600 python bench.py --workload groupby_minmax --gb-algo 1 --no-cpu-baseline ) > $O/bench13_minmax_global.jsonl 2>> $L
( time timeout 900 python bench.py --no-cpu-baseline --through-cpp ) > $O/bench13_through_cpp.jsonl 2>> $L
CUDF_AMD_ALLOC=plain timeout 600 tests/cpp/cudf_api_bench 1e9 3 1 > $O/bench13_cpp_plain_alloc.json 2>> $L
python - <<'PY'
import json
for f in ('minmax', 'minmax_global'):
    try:
        d = json.loads(open(f'gpurun_out/bench13_{f}.jsonl').read().strip().split('\n')[-1])
        print(f, round(d['ms_per_step'], 3), 'ms', round(d['roofline']['frac'], 3))
    except Exception as e:
        print(f, 'failed', e)
try:
    d = json.loads(open('gpurun_out/bench13_through_cpp.jsonl').read().strip().split('\n')[-1])
    print('c-abi sort', round(d['ms_per_step'], 2), 'join', round(d['join']['ms_per_step'], 2), 'groupby', round(d['groupby']['ms_per_step'], 2))
    print('through_cpp', d['through_cpp'])
except Exception as e:
    print('through_cpp failed', e)
print('plain alloc', open('gpurun_out/bench13_cpp_plain_alloc.json').read())
PY
tail -5 $L
