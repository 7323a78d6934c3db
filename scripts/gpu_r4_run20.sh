#!/bin/bash
# round 4, run 20: the overflow flag raised per thread behind a load (no extra barrier reduction in k_hf_scatter) -- cursor-path and
# big-cell tests, headline + the lines that reach level 1 with overflowing cells
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out; R=20; L=$O/r4_run20.log; : > $L
timeout 900 python -m pytest tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py -q -x 2>&1 | tail -5 > $O/r4_run${R}_tests.log
rb() { local tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r4_run${R}_bench_${tag}.jsonl; }
rb sort --workload sort --steps 10
rb sort_b --workload sort --steps 10
rb sort_signed_range --workload sort --key-range -1000000000000 1000000000000
rb sort_hot1e8 --workload sort --hot-copies 1e8
rb sort_hot4e8 --workload sort --hot-copies 4e8
rb sort_normal --workload sort --key-dist normal
cat $O/r4_run${R}_tests.log
python - <<PY | tee $O/r4_run${R}_sort_lines.txt
import json, glob
print("# round 4 run $R: python bench.py --workload sort (1e9 int64 rows)")
for f in sorted(glob.glob("$O/r4_run${R}_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("_bench_")[1][:-6], "|", d["config"]["workload"], "|", round(d["ms_per_step"], 3), "ms |", {k: si.get(k) for k in ("shift0", "bits2", "max_cell", "lsd_passes", "cursor_path_state", "big_cells")})
    except Exception as e:
        print(f, "unreadable", e)
PY
