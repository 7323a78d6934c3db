#!/bin/bash
# round 5 run 13+: the warp of the splitter mode's cell maps + the per-sign LUT of float keys: sort tests, then the distributions
set -u
R=${1:-13}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
( timeout 1200 python -m pytest tests/test_gpu_sort_splitters.py tests/test_gpu_sort_float_cursor.py tests/test_gpu_sort_counting.py tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py -m gpu -q -x 2>&1 | tail -30 ) > $O/r5_run${R}_tests.log
tail -n 8 $O/r5_run${R}_tests.log
rb() { local tag=$1; shift; timeout 300 python bench.py --workload sort --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r5_run${R}_bench_sort_${tag}.jsonl; }
rb f64_normal --key-type float64 --key-dist normal
rb f64_uniform --key-type float64 --key-dist uniform
rb clusters --key-dist clusters
rb normal --key-dist normal
rb zipf --key-dist zipf
rb lognormal --key-dist lognormal
rb uniform
python - $R <<'PY' | tee $O/r5_run${R}_sort_lines.txt
import json, glob, sys
R = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/r5_run{R}_bench_sort_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d.get("roofline") or {}
        si = r.get("sort_info") or {}
        print(f.split("_bench_sort_")[1][:-6], "|", round(d["ms_per_step"], 3), "ms |", {k: si.get(k) for k in ("bits2", "lsd_passes", "cursor_path_state", "big_cells", "splitters")}, {k[:22]: round(v, 2) for k, v in (r.get("kernels_ms") or {}).items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -E "Error|error|Traceback|assert" $L | head -20
