#!/bin/bash
# round 3 run 17: groupby on several key columns in one partition pass (gx_groupby_sum_count_wide): parity + bench
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run17.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_groupby_wide.py tests/test_gpu_dataframe.py tests/test_gpu_join_kinds_multikey.py -q -x > $O/r3_run17_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -25 $O/r3_run17_pytest.log | tee -a $L
timeout 300 python bench.py --workload groupby_multikey --steps 5 --warmup 2 --no-cpu 2>>$L | tail -1 > $O/r3_run17_bench_groupby_multikey.jsonl
python - <<'PY' | tee -a $L
import json
for l in open('gpurun_out/r3_run17_bench_groupby_multikey.jsonl'):
    try: d=json.loads(l)
    except Exception: print('bad line', l[:300]); continue
    print(d['config'].get('workload','')[:90], round(d['ms_per_step'],3))
PY
tail -5 $L
echo finished | tee -a $L
