#!/bin/bash
# round 5 run 2: the counting sort of narrow key ranges (tests + 1e9-row lines), the tag probe in its occupancy form (knob 4 / 5:
# tests, A/B lines against the pipelined probe, SQ counters)
set -u
R=${1:-2}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_sort_counting.py -m gpu -q -x 2>&1 | tail -15 ) > $O/r5_run${R}_tests_counting.log
echo "tests counting done $(( $(date +%s) - t0 )) s" | tee -a $L
( timeout 600 python -m pytest tests/test_gpu_join_partition_modes.py -m gpu -q -x -k "l2_resident" 2>&1 | tail -15 ) > $O/r5_run${R}_tests_probe.log
echo "tests probe done $(( $(date +%s) - t0 )) s" | tee -a $L
for k in 0 4 5; do
  timeout 300 python bench.py --workload join --no-cpu-baseline --join-probe-kernel $k 2>> $L | tail -1 > $O/r5_run${R}_bench_join_k$k.jsonl
done
echo "join ab done $(( $(date +%s) - t0 )) s" | tee -a $L
timeout 300 python bench.py --workload sort --no-cpu-baseline --key-range 100 10001 2>> $L | tail -1 > $O/r5_run${R}_bench_sort_range_100_10001.jsonl
timeout 300 python bench.py --workload sort --no-cpu-baseline --key-range 0 30000 2>> $L | tail -1 > $O/r5_run${R}_bench_sort_range_0_30000.jsonl
timeout 300 python bench.py --workload sort --no-cpu-baseline 2>> $L | tail -1 > $O/r5_run${R}_bench_sort.jsonl
echo "sort lines done $(( $(date +%s) - t0 )) s" | tee -a $L
bash scripts/gpu_pmc_sq.sh join k_pj --join-probe-kernel 4 > /dev/null 2>&1
cp $O/pmc_sq_join_summary.txt $O/r5_run${R}_pmc_sq_join_k4.txt
echo "pmc done $(( $(date +%s) - t0 )) s" | tee -a $L
python - $R <<'PY' | tee -a $L
import json, glob, sys
R = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/r5_run{R}_bench_*.jsonl")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        si = r.get("sort_info") or {}
        print(f.split("_bench_")[1], round(d["ms_per_step"], 3), "ms", {k[:20]: round(v, 2) for k, v in (r.get("kernels_ms") or {}).items()}, "build", d.get("join_build_ms"), "state", si.get("cursor_path_state"), "lsd", si.get("lsd_passes"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -n 6 $O/r5_run${R}_tests_counting.log
tail -n 6 $O/r5_run${R}_tests_probe.log
grep -A17 "k_pj4_probe_tags.*dispatch 224" $O/r5_run${R}_pmc_sq_join_k4.txt | head -20
grep -E "Error|error|Traceback|assert" $L | head -20
