#!/bin/bash
# round 3 run 16: groupby with 512 partitions (bins in registers, keys first) A/B on dense / sparse keys; sorted_order without the key write-back
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run16.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_sort.py "tests/test_gpu_parity_1e8.py" tests/test_gpu_dataframe.py -q -x > $O/r3_run16_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/r3_run16_pytest.log | tee -a $L
: > $O/r3_run16_bench_groupby_ab.jsonl
for keys in dense random random64; do
  for pb in 9 8; do
    timeout 300 python bench.py --workload groupby --steps 10 --warmup 3 --no-cpu --gb-keys $keys --gb-pbits $pb 2>>$L | tail -1 >> $O/r3_run16_bench_groupby_ab.jsonl
  done
done
timeout 300 python bench.py --workload groupby_minmax --steps 10 --warmup 3 --no-cpu 2>>$L | tail -1 >> $O/r3_run16_bench_groupby_ab.jsonl
timeout 300 python bench.py --workload sorted_order --steps 5 --warmup 2 --no-cpu 2>>$L | tail -1 > $O/r3_run16_bench_sorted_order.jsonl
python - <<'PY' | tee -a $L
import json
for f in ('gpurun_out/r3_run16_bench_groupby_ab.jsonl','gpurun_out/r3_run16_bench_sorted_order.jsonl'):
    for l in open(f):
        try: d=json.loads(l)
        except Exception: print('bad line', l[:300]); continue
        print(d['config'].get('workload','')[:70], round(d['ms_per_step'],3), {k:v for k,v in d['config'].items() if k in ('gb_spec','gb_pbits','gb_keys')})
PY
echo finished | tee -a $L
