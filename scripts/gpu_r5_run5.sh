#!/bin/bash
# round 5 run 5+: the splitter mode of the sort -- tests, then 1e9-row lines on the distributions that used to be declined
set -u
R=${1:-5}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_sort_splitters.py -m gpu -q ${PYTEST_X:--x} 2>&1 | tail -40 ) > $O/r5_run${R}_tests_split.log
echo "tests split done $(( $(date +%s) - t0 )) s" | tee -a $L
tail -n 30 $O/r5_run${R}_tests_split.log
if [ "${SKIP_OTHER_TESTS:-0}" != "1" ]; then
( timeout 900 python -m pytest tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_big_cells.py tests/test_gpu_sort_counting.py tests/test_gpu_sort_place.py -m gpu -q -x 2>&1 | tail -15 ) > $O/r5_run${R}_tests_sort.log
echo "tests sort done $(( $(date +%s) - t0 )) s" | tee -a $L
tail -n 6 $O/r5_run${R}_tests_sort.log
fi
rb() { local tag=$1; shift; timeout 300 python bench.py --workload sort --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r5_run${R}_bench_sort_${tag}.jsonl; }
rb uniform
rb normal --key-dist normal
rb zipf --key-dist zipf
rb lognormal --key-dist lognormal
rb clusters --key-dist clusters
rb sorted --key-dist sorted
rb signed_range --key-range -1000000000000 1000000000000
rb hot1e8 --hot-copies 1e8
echo "sort lines done $(( $(date +%s) - t0 )) s" | tee -a $L
python - $R <<'PY' | tee $O/r5_run${R}_sort_robustness.txt
import json, glob, sys
R = sys.argv[1]
print(f"# round 5 run {R}: python bench.py --workload sort --steps 3 --warmup 1 on key distributions other than uniform 64-bit (1e9 int64 rows)")
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
