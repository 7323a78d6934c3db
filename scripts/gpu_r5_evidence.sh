#!/bin/bash
# round 5 evidence run: usage gpu_r5_evidence.sh <run number> [pmc] [tests] [robust]
#   default bench line (with the CPU legs), rocprofv3 --kernel-trace --stats summary of the same command, sorted_order line;
#   with "pmc": FETCH_SIZE / WRITE_SIZE passes (separate runs) of sort / join / groupby at 1e9 rows -> r5_pmc_traffic_1e9.json
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
R=${1:-99}
mkdir -p gpurun_out
O=gpurun_out
L=$O/r5_run$R.log
: > $L
( time timeout 900 python bench.py ) > $O/r5_run${R}_bench_default.jsonl 2>> $L
timeout 300 python bench.py --workload sorted_order --no-cpu-baseline > $O/r5_run${R}_bench_sorted_order.jsonl 2>> $L
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_default" -o default -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-robustness --no-through-cpp) > $O/r5_run${R}_bench_under_rocprof.jsonl 2>> $L
db=$(find $O/prof_default -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 5 run $R: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-robustness --no-through-cpp (sort + sorted_order + join + groupby, 5 steps + 2 warm-up each)" | head -60 | cut -c1-190 > $O/r5_run${R}_default_kernel_stats.txt
find $O/prof_default -name "*.db" -delete
if [ "${2:-}" = "pmc" ]; then
  pmc() { # workload, counter
    local wl=$1; local ctr=$2
    local lc=$(echo $ctr | tr 'A-Z' 'a-z')
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_${wl}_${lc}" -o $wl --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline) >> $L 2>&1
  }
  for wl in sort sorted_order join groupby; do
    pmc $wl FETCH_SIZE
    pmc $wl WRITE_SIZE
  done
  python scripts/pmc_to_json.py $O $O/r5_pmc_traffic_1e9.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of python bench.py --workload <w> --rows 1e9 --steps 1 --warmup 0 (scripts/gpu_r5_evidence.sh $R pmc)" | tee $O/r5_run${R}_pmc_traffic.txt
  find $O/pmc_* -name "*.csv" -size +1M -delete
fi
if [ "${3:-}" = "tests" ]; then
  timeout 400 python -m pytest tests/test_gpu_cpp_parity.py tests/test_gpu_sort_place.py -m gpu -q -x -k "not 70000000 and not capacity and not float64" > $O/r5_run${R}_pytest.log 2>&1
  echo "pytest exit $?" | tee -a $L
  tail -4 $O/r5_run${R}_pytest.log | tee -a $L
fi
if [ "${4:-}" = "robust_separately" ]; then
  # value distributions the sort is NOT tuned for (cost must not depend on the distribution): measured, whatever they cost
  rb() { local tag=$1; shift; timeout 300 python bench.py --workload sort --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>> $L | tail -1 > $O/r5_run${R}_bench_sort_${tag}.jsonl; }
  rb normal --key-dist normal
  rb zipf --key-dist zipf
  rb sorted --key-dist sorted
  rb signed_range --key-range -1000000000000 1000000000000
  rb range_100_10001 --key-range 100 10001
  python - <<PY | tee $O/r5_run${R}_sort_robustness.txt
import json, glob
print("# round 5 run $R: python bench.py --workload sort --steps 3 on key distributions other than uniform 64-bit (1e9 int64 rows)")
for f in sorted(glob.glob("$O/r5_run${R}_bench_sort_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        si = (d.get("roofline") or {}).get("sort_info") or {}
        print(f.split("_bench_sort_")[1][:-6], "|", d["config"]["workload"], "|", round(d["ms_per_step"], 3), "ms |", {k: si.get(k) for k in ("bits2", "max_cell", "lsd_passes", "cursor_path_state", "big_cells")})
    except Exception as e:
        print(f, "unreadable", e)
PY
fi
python - <<PY | tee -a $L
import json
for f in ('$O/r5_run${R}_bench_default.jsonl', '$O/r5_run${R}_bench_sorted_order.jsonl'):
    for line in open(f):
        try: d = json.loads(line)
        except Exception: continue
        r = d.get('roofline') or {}
        print(d['config']['workload'][:60], round(d['ms_per_step'], 3), 'ms | frac', round(r.get('frac', 0), 3), '| path_frac', round(r.get('path_frac', 0), 3), {k[:22]: round(v, 2) for k, v in (r.get('kernels_ms') or {}).items()})
        for k in ('join', 'groupby'):
            if k in d:
                rr = d[k].get('roofline') or {}
                print('  ', k, round(d[k]['ms_per_step'], 3), 'ms | frac', round(rr.get('frac', 0), 3), {kk[:18]: round(v, 2) for kk, v in (rr.get('kernels_ms') or {}).items()})
PY
head -24 $O/r5_run${R}_default_kernel_stats.txt | cut -c1-170
grep -E "real|Error|error|Traceback" $L | head
