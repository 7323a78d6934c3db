#!/bin/bash
# round 4, run 5: big-cell polish (extra level-1 bit needs 4 full buckets, k_big_plan per bucket, two-tier lds_rank) -- tests + the
# three sort lines again; groupby A/B of the partition bits on dense and sparse keys (same box)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_sort_big_cells.py tests/test_gpu_sort_cursor_path.py -x -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_join_groupby.py -x -q 2>&1 | tail -4
} > $O/r4_run5_tests.log 2>&1
for hc in 0 1e6 1e8; do
  python bench.py --workload sort --no-cpu-baseline --steps 5 --hot-copies $hc > $O/r4_run5_bench_sort_hot_$hc.jsonl 2> $O/r4_run5_err_sort_$hc.txt
done
python bench.py --workload sort --no-cpu-baseline --steps 5 --key-range 0 1000000000000 > $O/r4_run5_bench_sort_range1e12.jsonl 2> $O/r4_run5_err_sort_range.txt
for pb in 8 9; do for keys in dense random; do
  python bench.py --workload groupby --no-cpu-baseline --steps 5 --gb-pbits $pb --gb-keys $keys > $O/r4_run5_bench_groupby_pbits${pb}_${keys}.jsonl 2> $O/r4_run5_err_gb.txt
done; done
cat $O/r4_run5_tests.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_run5_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("config") or {}).get("sort_info") or {}
        print(f.split("r4_run5_bench_")[1], round(d["ms_per_step"], 3), {k: si.get(k) for k in ("bits2", "cursor_path_state", "big_cells")} if si else "")
    except Exception as e:
        print(f, "unreadable", e)
PY
