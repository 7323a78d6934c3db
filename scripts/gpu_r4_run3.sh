#!/bin/bash
# round 4, run 3: big cells of the cursor-path sort (hot values) under their tests + the existing cursor / place tests,
# the sort bench with 0 / 1e6 / 1e8 copies of one value, and a kernel-trace of the join bench (build breakdown)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_sort_big_cells.py -x -q --durations=5 2>&1 | tail -25
timeout 900 python -m pytest tests/test_gpu_sort_cursor_path.py tests/test_gpu_sort_place.py -x -q 2>&1 | tail -5
} > gpurun_out/r4_run3_tests.log 2>&1
for hc in 0 1e6 1e8; do
  python bench.py --workload sort --no-cpu-baseline --steps 5 --hot-copies $hc > gpurun_out/r4_run3_bench_sort_hot_$hc.jsonl 2> gpurun_out/r4_run3_bench_sort_hot_$hc.err
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r4_run3_prof_join -o join -- python bench.py --workload join --no-cpu-baseline --steps 3 > gpurun_out/r4_run3_prof_join.log 2>&1
f=$(find gpurun_out/r4_run3_prof_join -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -25 "$f" > gpurun_out/r4_run3_join_kernel_stats.txt
rm -rf gpurun_out/r4_run3_prof_join
tail -30 gpurun_out/r4_run3_tests.log
python - <<'PY'
import json
for hc in ("0", "1e6", "1e8"):
    f = f"gpurun_out/r4_run3_bench_sort_hot_{hc}.jsonl"
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(hc, d.get("ms_per_step"), d.get("config", {}).get("sort_info") or d.get("sort_info"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".jsonl", ".err")).read()[-600:])
PY
cat gpurun_out/r4_run3_join_kernel_stats.txt | cut -c1-150
