#!/bin/bash
# round 2 run 20: msd pass with the first look-back window prefetched; hash_join build trace through the C++ API
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run20.log
: > $L
( time timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_parity_1e8.py -m gpu -q -k "sort" ) > $O/pytest_gpu20.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu20.log | head | tee -a $L
for i in 1 2; do ( timeout 600 python bench.py --workload sort --no-cpu-baseline ) >> $O/bench20_sort.jsonl 2>> $L; done
( timeout 600 python bench.py --workload sorted_order --no-cpu-baseline ) >> $O/bench20_sort.jsonl 2>> $L
CUDF_API_BENCH_TRACE=1 timeout 600 tests/cpp/cudf_api_bench 1e9 3 1 > $O/bench20_cpp.json 2> $O/bench20_cpp_trace.txt
python - <<'PY'
import json
for line in open('gpurun_out/bench20_sort.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    r = d['roofline']
    print(d['config']['workload'][:50], round(d['ms_per_step'], 2), 'ms', 'hist', round(r.get('hist_kernel_ms', 0), 2), [round(v, 2) for v in (r.get('kernels_ms') or {}).values()])
PY
cat $O/bench20_cpp_trace.txt | grep -v amdgpu; cat $O/bench20_cpp.json
