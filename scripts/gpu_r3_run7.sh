#!/bin/bash
# round 3 run 7: speculative (one-read) exchange partition passes in gxd + the groupby's hist-free partition pass: tests, then
# the per-rank step costs (xp_distributed_single_rank) and the groupby A/B
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run7.log
: > $L
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_join_groupby.py -q --durations=5 > $O/r3_run7_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/r3_run7_pytest.log | tee -a $L
timeout 300 tests/cpp/cudf_api_tests > $O/r3_run7_cpp.log 2>&1
echo "cpp tests exit $?" | tee -a $L
tail -3 $O/r3_run7_cpp.log | tee -a $L
: > $O/r3_run7_bench_groupby.jsonl
for gs in 1 0; do
  echo "== groupby spec=$gs" | tee -a $L
  timeout 600 python bench.py --workload groupby --no-cpu-baseline --gb-spec $gs >> $O/r3_run7_bench_groupby.jsonl 2>> $L
  timeout 600 python bench.py --workload groupby_minmax --no-cpu-baseline --gb-spec $gs >> $O/r3_run7_bench_groupby.jsonl 2>> $L
done
python - <<'PY' | tee -a gpurun_out/r3_run7.log
import json
for line in open('gpurun_out/r3_run7_bench_groupby.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    print(d['config']['workload'][:70], d['config'].get('gb_spec'), round(d['ms_per_step'], 3), 'ms')
PY
timeout 900 python scripts/xp/xp_distributed_single_rank.py 1e9 > $O/r3_xp_distributed_single_rank.txt 2>> $L
cat $O/r3_xp_distributed_single_rank.txt | tee -a $L
