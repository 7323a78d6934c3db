#!/bin/bash
# round 5 run 4: join probe with the tag windows read from the L2 (knob 6 / 7: 256 partitions of 2^20 slots) against the LDS-tag kernels
set -u
R=${1:-4}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_join_partition_modes.py -m gpu -q -x -k "l2_resident and (6 or 7)" 2>&1 | tail -15 ) > $O/r5_run${R}_tests_probe.log
echo "tests probe done $(( $(date +%s) - t0 )) s" | tee -a $L
for k in ${KERNELS:-0 5 6 7}; do
  timeout 300 python bench.py --workload join --no-cpu-baseline --join-probe-kernel $k 2>> $L | tail -1 > $O/r5_run${R}_bench_join_k$k.jsonl
done
echo "join ab done $(( $(date +%s) - t0 )) s" | tee -a $L
for k in ${PMC_KERNELS:-6}; do
  bash scripts/gpu_pmc_sq.sh join k_pj --join-probe-kernel $k > /dev/null 2>&1
  cp $O/pmc_sq_join_summary.txt $O/r5_run${R}_pmc_sq_join_k$k.txt
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$GRAFT_REPO_ROOT/$O/pmc_tcc_k$k" -o tcc --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload join --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline --join-probe-kernel $k) >> $L 2>&1
  python - $(find $O/pmc_tcc_k$k -name "*counter_collection.csv") > $O/r5_run${R}_pmc_tcc_join_k$k.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(dict)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if "k_pj" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:60], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for (k, i), v in sorted(d.items()):
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    if h + m > 1e6:
        print("%-62s dispatch %-4s hit %14.6g miss %14.6g hit rate %.3f" % (k, i, h, m, h / (h + m) if h + m else 0))
PY
  find $O/pmc_tcc_k$k -name "*.csv" -size +1M -delete
done
echo "pmc done $(( $(date +%s) - t0 )) s" | tee -a $L
python - $R <<'PY' | tee -a $L
import json, glob, sys
R = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/r5_run{R}_bench_*.jsonl")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("_bench_")[1], round(d["ms_per_step"], 3), "ms", {k[:20]: round(v, 2) for k, v in (r.get("kernels_ms") or {}).items()}, "build", d.get("join_build_ms"), "pbits", d.get("join_partition_bits"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -n 6 $O/r5_run${R}_tests_probe.log
for k in ${PMC_KERNELS:-6}; do grep -A17 "k_pj4_probe_tags.*dispatch 224\|k_pj2_scatter.*dispatch 222" $O/r5_run${R}_pmc_sq_join_k$k.txt | head -40; cat $O/r5_run${R}_pmc_tcc_join_k$k.txt; done
grep -E "Error|error|Traceback|assert" $L | head -20
