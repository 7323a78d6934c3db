#!/bin/bash
# round 3 run 12: cursor path polish (speculative top-byte sample histogram, gated status clear): tests + bench
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run12.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort.py -q -x > $O/r3_run12_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/r3_run12_pytest.log | tee -a $L
timeout 300 python bench.py --workload sort --steps 20 --warmup 3 --no-cpu 2>>$L | tail -1 > $O/r3_run12_bench_sort.jsonl
timeout 300 python bench.py --workload sort --steps 10 --warmup 3 --no-cpu --key-range 100 10000 2>>$L | tail -1 >> $O/r3_run12_bench_sort.jsonl
python - <<'PY' | tee -a $L
import json
for l in open('gpurun_out/r3_run12_bench_sort.jsonl'):
    try: d=json.loads(l)
    except Exception: print('bad line', l[:200]); continue
    r=d.get('roofline') or {}
    print(d['config'].get('workload','')[:60], d['ms_per_step'], {k[:24]:round(v,3) for k,v in (r.get('kernels_ms') or {}).items()}, 'hist', r.get('hist_kernel_ms'), (r.get('sort_info') or {}))
PY
echo finished | tee -a $L
