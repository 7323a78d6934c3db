#!/bin/bash
# round 4, run 13: per-bucket cell slots on the cursor path only (neighbour-smoothed, budgeted), look-back fixed -- every sort test + the sharded-sort loopback tests, then the sort lines
# (uniform, hot values, [0, 1e12), 1.25e9 rows, sorted_order, int32) for regressions
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "sort" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity_1e8.py tests/test_gpu_cpp_parity.py -x -q -k "sort or rank or top_k or segmented or scan" 2>&1 | tail -3
} > $O/r4_run13_tests.log 2>&1
b() { python bench.py --no-cpu-baseline --steps 5 "$@" 2>> $O/r4_run13_err.txt | tail -1; }
b --workload sort > $O/r4_run13_bench_sort.jsonl
b --workload sort --hot-copies 1e6 > $O/r4_run13_bench_sort_hot1e6.jsonl
b --workload sort --hot-copies 1e8 > $O/r4_run13_bench_sort_hot1e8.jsonl
b --workload sort --key-range 0 1000000000000 > $O/r4_run13_bench_sort_range1e12.jsonl
b --workload sort --rows 1.25e9 > $O/r4_run13_bench_sort_1p25e9.jsonl
b --workload sorted_order > $O/r4_run13_bench_sorted_order.jsonl
b --workload sorted_order --key-range 0 1000000000000 > $O/r4_run13_bench_sorted_order_range1e12.jsonl
cat $O/r4_run13_tests.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_run13_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("r4_run13_bench_")[1], round(d["ms_per_step"], 3), {k: si.get(k) for k in ("bits2", "max_cell", "lsd_passes", "cursor_path_state", "big_cells")})
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $O/r4_run13_err.txt
