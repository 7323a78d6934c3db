#!/bin/bash
# round 4, run 16: early decline of value distributions the two levels cannot split (k_hf_plan stages 1 / 2, k_hy_plan stage 1),
# k_hist_all's popular-digit aggregation -- sort tests, then the robustness lines of run 14 again + the headline lines for regressions
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out; R=16; L=$O/r4_run16.log; : > $L
{
timeout 1200 python -m pytest tests/test_gpu_sort.py tests/test_gpu_sort_place.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "sort" 2>&1 | tail -3
} > $O/r4_run${R}_tests.log 2>&1
rb() { local tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r4_run${R}_bench_${tag}.jsonl; }
rb sort --workload sort --steps 5
rb sort_hot1e6 --workload sort --hot-copies 1e6
rb sort_hot1e8 --workload sort --hot-copies 1e8
rb sort_normal --workload sort --key-dist normal
rb sort_zipf --workload sort --key-dist zipf
rb sort_sorted --workload sort --key-dist sorted
rb sort_signed_range --workload sort --key-range -1000000000000 1000000000000
rb sorted_order --workload sorted_order
rb sorted_order_normal --workload sorted_order --key-dist normal
cat $O/r4_run${R}_tests.log
python - <<PY | tee $O/r4_run${R}_sort_robustness.txt
import json, glob
print("# round 4 run $R: python bench.py --workload sort|sorted_order --steps 3 on key distributions other than uniform 64-bit (1e9 int64 rows), after the early decline")
for f in sorted(glob.glob("$O/r4_run${R}_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("_bench_")[1][:-6], "|", d["config"]["workload"], "|", round(d["ms_per_step"], 3), "ms |", {k: si.get(k) for k in ("bits2", "max_cell", "lsd_passes", "cursor_path_state", "big_cells")})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -v amdgpu.ids $L | tail -5
