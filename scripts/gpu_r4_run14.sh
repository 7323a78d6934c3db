#!/bin/bash
# round 4, run 14: the sort on value distributions it is not tuned for (normal, Zipf-like, already sorted, a range that crosses zero,
# the reference benchmark's [100, 10001)) -- measured, whatever they cost
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out; R=14; L=$O/r4_run14.log; : > $L
rb() { local tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r4_run${R}_bench_${tag}.jsonl; }
rb sort_normal --workload sort --key-dist normal
rb sort_zipf --workload sort --key-dist zipf
rb sort_sorted --workload sort --key-dist sorted
rb sort_signed_range --workload sort --key-range -1000000000000 1000000000000
rb sort_range_100_10001 --workload sort --key-range 100 10001
rb sorted_order_normal --workload sorted_order --key-dist normal
python - <<PY | tee $O/r4_run${R}_sort_robustness.txt
import json, glob
print("# round 4 run $R: python bench.py --workload sort|sorted_order --steps 3 on key distributions other than uniform 64-bit (1e9 int64 rows)")
for f in sorted(glob.glob("$O/r4_run${R}_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("_bench_")[1][:-6], "|", d["config"]["workload"], "|", round(d["ms_per_step"], 3), "ms |", {k: si.get(k) for k in ("bits2", "max_cell", "lsd_passes", "cursor_path_state", "big_cells")})
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 $L
