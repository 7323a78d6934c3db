#!/bin/bash
# round 2 run 3: sort tests (float gating fixed), null-key joins, C++ API cases, 1e8-row parity, SQ counters of the sort
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run3.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_sort.py tests/test_gpu_join_kinds_multikey.py tests/test_cpp_api.py tests/test_gpu_parity_1e8.py -m gpu -q --durations=8 > $O/pytest_gpu3.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|CHECK failed|FAIL\]" $O/pytest_gpu3.log | head -30 | tee -a $L
grep -A10 "slowest" $O/pytest_gpu3.log | tee -a $L
timeout 300 python bench.py --workload sort --no-cpu-baseline --rows 1.05e9 > $O/bench3_sort_105.jsonl 2>> $L
python -c "
import json; d=json.loads(open('gpurun_out/bench3_sort_105.jsonl').read()); r=d['roofline']; print(round(d['ms_per_step'],2), r['sort_info'], [round(v,2) for v in r.get('kernels_ms',{}).values()])"
bash scripts/gpu_pmc_sq.sh sort gx::sort > /dev/null 2>&1
grep -A17 "k_local_sort\|k_msd_pass" $O/pmc_sq_sort_summary.txt | head -120
