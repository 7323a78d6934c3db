#!/bin/bash
# round 5 run 16: level 0 with one instantiation per LUT form; SQ counters of the cell stage / level 1 in splitter mode against the bit digits
set -u
R=${1:-16}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
( timeout 900 python -m pytest tests/test_gpu_sort_splitters.py tests/test_gpu_sort_float_cursor.py -m gpu -q -x 2>&1 | tail -30 ) > $O/r5_run${R}_tests.log
tail -n 4 $O/r5_run${R}_tests.log
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
pmc() { # tag, counters, bench args
  local name=$1; shift
  local ctr=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload sort --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline "$@") >> $L 2>&1
}
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
pmc r16_uni1 "$C1"
pmc r16_nrm1 "$C1" --key-dist normal
pmc r16_uni2 "$C2"
pmc r16_nrm2 "$C2" --key-dist normal
python - $(find $O/pmc_r16_* -name "*counter_collection.csv") > $O/r5_run${R}_pmc_sq_sort_split_vs_bits.txt <<'PY'
import collections, csv, sys
files = sys.argv[1:]
d = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    tag = "normal (splitter mode)" if "_nrm" in f else "uniform (bit digits)"
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("k_local_place", "k_hf_scatter", "k_sp_level0", "k_local_sort")):
            continue
        key = (tag, k[:96])
        d[key][r["Counter_Name"]] += float(r["Counter_Value"])
print("# round 5 run 16: SQ counters summed over the dispatches of one 1e9-row sort (bench.py --workload sort --rows 1e9 --steps 1 --warmup 0), rocprofv3 --kernel-trace --pmc, two passes")
for (tag, k), v in sorted(d.items()):
    if v.get("SQ_WAVE_CYCLES", 0) + v.get("SQ_BUSY_CYCLES", 0) < 1e6:
        continue
    print(tag, "|", k)
    for a, b in sorted(v.items()):
        print("    %-22s %14.6g" % (a, b))
PY
find $O/pmc_r16_* -name "*.csv" -size +1M -delete
cat $O/r5_run${R}_pmc_sq_sort_split_vs_bits.txt | head -150
grep -E "Error|error|Traceback|assert" $L | head -20
