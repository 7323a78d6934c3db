#!/bin/bash
# round 3 run 10: cursor path of the hybrid sort (sampled plan, atomic-cursor partition levels): parity, A/B bench
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run10.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort.py tests/test_gpu_parity_1e8.py::test_sort_1e8_matches_c_oracle -q -x > $O/r3_run10_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/r3_run10_pytest.log | tee -a $L
for extra in "" "--no-cursor"; do
  timeout 300 python bench.py --workload sort --steps 10 --warmup 3 --no-cpu $extra 2>>$L | tail -1 >> $O/r3_run10_bench_sort_ab.jsonl
done
timeout 300 python bench.py --workload sort --rows 1.1e9 --steps 5 --warmup 2 --no-cpu 2>>$L | tail -1 >> $O/r3_run10_bench_sort_ab.jsonl
timeout 300 python bench.py --workload sort --rows 1.1e9 --steps 5 --warmup 2 --no-cpu --no-cursor 2>>$L | tail -1 >> $O/r3_run10_bench_sort_ab.jsonl
python - <<'PY' | tee -a $L
import json
for l in open('gpurun_out/r3_run10_bench_sort_ab.jsonl'):
    try: d=json.loads(l)
    except Exception: print('bad line', l[:200]); continue
    r=d.get('roofline') or {}
    print(d['config'].get('workload','')[:40], d['ms_per_step'], {k[:24]:round(v,3) for k,v in (r.get('kernels_ms') or {}).items()}, 'hist', r.get('hist_kernel_ms'), (r.get('sort_info') or {}))
PY
echo finished | tee -a $L
