#!/bin/bash
# round 4, run 30: the cell kernels request a cell's size, output position and slot together -- placement tests, the sort and
# sorted_order lines, then the PMC passes for the sources as they are
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
O=gpurun_out; L=$O/r4_run30.log; : > $L
timeout 100 python -m pytest tests/test_gpu_sort_place.py tests/test_gpu_sort_big_cells.py -q -x 2>&1 | tail -3 > $O/r4_run30_tests.log
cat $O/r4_run30_tests.log
timeout 60 python bench.py --workload sort --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/r4_run30_bench_sort.jsonl
timeout 60 python bench.py --workload sorted_order --no-cpu-baseline --steps 5 2>/dev/null | tail -1 > $O/r4_run30_bench_sorted_order.jsonl
python - <<'PY'
import json
for w in ("sort", "sorted_order"):
    d = json.loads(open(f"gpurun_out/r4_run30_bench_{w}.jsonl").read()); r = d["roofline"]
    print(w, round(d["ms_per_step"], 3), "frac", round(r["frac"], 3), {k[:22]: round(v, 2) for k, v in r["kernels_ms"].items()})
PY
pmc() { local wl=$1; local ctr=$2; local lc=$(echo $ctr | tr 'A-Z' 'a-z')
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_${wl}_${lc}" -o $wl --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline) >> $L 2>&1
}
for wl in sort sorted_order join groupby; do
  pmc $wl FETCH_SIZE
  pmc $wl WRITE_SIZE
done
python scripts/pmc_to_json.py $O $O/r4_pmc_traffic_1e9.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of python bench.py --workload <w> --rows 1e9 --steps 1 --warmup 0 (scripts/gpu_r4_run30.sh)" | tee $O/r4_run30_pmc_traffic.txt | grep GROUP | cut -c1-90
find $O/pmc_* -name "*.csv" -size +1M -delete
