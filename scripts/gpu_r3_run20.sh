#!/bin/bash
# round 3 run 20: sampled slot capacities in the single-key groupby too; C++ wide-groupby case; benches
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run20.log
: > $L
timeout 600 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_groupby_wide.py tests/test_gpu_dataframe.py "tests/test_gpu_parity_1e8.py::test_groupby_1e8_matches_c_oracle" -q -x --timeout 120 > $O/r3_run20_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -6 $O/r3_run20_pytest.log | cut -c1-200 | tee -a $L
(timeout 300 tests/cpp/cudf_api_tests 2>&1 | grep -v "^\[ OK \]" | tail -8) | tee -a $L
: > $O/r3_run20_bench_groupby.jsonl
for keys in dense random random64; do
  timeout 300 python bench.py --workload groupby --steps 10 --warmup 3 --no-cpu --gb-keys $keys 2>>$L | tail -1 >> $O/r3_run20_bench_groupby.jsonl
done
timeout 300 python bench.py --workload groupby_minmax --steps 10 --warmup 3 --no-cpu 2>>$L | tail -1 >> $O/r3_run20_bench_groupby.jsonl
timeout 300 python bench.py --workload groupby_multikey --steps 5 --warmup 2 --no-cpu 2>>$L | tail -1 >> $O/r3_run20_bench_groupby.jsonl
python - <<'PY' | tee -a $L
import json
for l in open('gpurun_out/r3_run20_bench_groupby.jsonl'):
    try: d=json.loads(l)
    except Exception: print('bad line', l[:300]); continue
    print(d['config'].get('workload','')[:90], round(d['ms_per_step'],3))
PY
echo finished | tee -a $L
