#!/bin/bash
# NOTE: runs 24 and 25 used a temporary host-side knob (environment variable GX_SORT_LAYOUT, read in sort_impl) that moved the two
# big buffers inside the scratch; it was removed again after these runs (no layout helped reproducibly).  Run 24 was this script
# with the layout list `0 1 2 3 4 0`.  Kept as the record of what was measured: profiles/r4_run24_layout_ab.txt, r4_run25_layout_ab.txt
# round 4, run 25 (EXPERIMENT): does the placement of the level-0 buffer and the cell buffer inside the scratch move level 0 / level 1?
# (level 0 3.84 -> 3.6 ms and level 1 3.63 -> 3.97 ms with the commit that shrank the cell buffer; no kernel code of either changed)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
for lay in 0 2 5 6 7 8 2 0; do
  GX_SORT_LAYOUT=$lay timeout 300 python bench.py --workload sort --no-cpu-baseline --steps 5 2>/dev/null | tail -1 > $O/r4_run25_bench_sort_layout${lay}_$RANDOM.jsonl
done
python - <<'PY' | tee gpurun_out/r4_run25_layout_ab.txt
import json, glob
print("# round 4 run 25: GX_SORT_LAYOUT A/B (0 current; 2 distance 2^34 as before the per-bucket slots; 5 / 6 / 7 distance rounded up to a multiple of 2^30 / 2^32 / 2^21 bytes; 8 half way to 2^34), 1e9 int64 keys")
for f in sorted(glob.glob("gpurun_out/r4_run25_bench_sort_layout*.jsonl")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f.split("layout")[1].split("_")[0], round(d["ms_per_step"], 3), {k[:22]: round(v, 2) for k, v in r["kernels_ms"].items()})
PY
