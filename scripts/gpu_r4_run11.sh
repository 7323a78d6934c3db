#!/bin/bash
# round 4, run 11: sorted_order's pairs switch to 16384-key cells on the device when the exact level-0 histogram shows fuller buckets
# (tests + the 1e9-row lines: full range = regression check, [0, 1e12) = the 91 ms cliff)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort.py -x -q 2>&1 | tail -5 > $O/r4_run11_tests.log
python bench.py --workload sorted_order --no-cpu-baseline --steps 5 > $O/r4_run11_bench_sorted_order.jsonl 2> $O/r4_run11_err.txt
python bench.py --workload sorted_order --no-cpu-baseline --steps 5 --key-range 0 1000000000000 > $O/r4_run11_bench_sorted_order_range1e12.jsonl 2>> $O/r4_run11_err.txt
cat $O/r4_run11_tests.log
python - <<'PY'
import json
for f in ("gpurun_out/r4_run11_bench_sorted_order.jsonl", "gpurun_out/r4_run11_bench_sorted_order_range1e12.jsonl"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("r4_run11_bench_")[1], round(d["ms_per_step"], 3), {k: si.get(k) for k in ("hybrid_used", "bits2", "max_cell", "lsd_passes")})
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $O/r4_run11_err.txt
