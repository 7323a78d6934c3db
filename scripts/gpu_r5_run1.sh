#!/bin/bash
# round 5 run 1: the round's new tests (pre-flight / sharded bench lines in a 1-rank group, the L2-resident direct probe, the
# sharded sort's unsampled outliers, the concurrent-probe C++ case), join probe A/B (tag probe vs the direct probe with 4 / 2 rows
# per thread), the verified 1e9-row sorted_order line, fresh SQ / TCC counters of today's join kernels
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run1.log
: > $L
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_join_partition_modes.py tests/test_gpu_bench_sharded.py -m gpu -q -x 2>&1 | tail -15 ) > $O/r5_run1_tests_a.log
echo "tests_a done $(( $(date +%s) - t0 )) s" | tee -a $L
( timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -m gpu -q -x -k "fused" 2>&1 | tail -15 ) > $O/r5_run1_tests_b.log
echo "tests_b done $(( $(date +%s) - t0 )) s" | tee -a $L
( timeout 600 ./tests/cpp/cudf_api_tests 2>&1 | tail -25 ) > $O/r5_run1_cpp_api.log
echo "cpp api done $(( $(date +%s) - t0 )) s" | tee -a $L
for k in 0 2 3; do
  timeout 300 python bench.py --workload join --no-cpu-baseline --join-probe-kernel $k 2>> $L | tail -1 > $O/r5_run1_bench_join_k$k.jsonl
done
echo "join ab done $(( $(date +%s) - t0 )) s" | tee -a $L
timeout 400 python bench.py --workload sorted_order --no-cpu-baseline 2>> $L | tail -1 > $O/r5_run1_bench_sorted_order.jsonl
echo "sorted_order done $(( $(date +%s) - t0 )) s" | tee -a $L
bash scripts/gpu_pmc_sq.sh join k_pj > /dev/null 2>&1
cp $O/pmc_sq_join_summary.txt $O/r5_run1_pmc_sq_join_k0.txt
bash scripts/gpu_pmc_sq.sh join k_pj --join-probe-kernel 2 > /dev/null 2>&1
cp $O/pmc_sq_join_summary.txt $O/r5_run1_pmc_sq_join_k2.txt
for k in 0 2; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$GRAFT_REPO_ROOT/$O/pmc_tcc_k$k" -o tcc --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload join --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline --join-probe-kernel $k) >> $L 2>&1
  python - $(find $O/pmc_tcc_k$k -name "*counter_collection.csv") > $O/r5_run1_pmc_tcc_join_k$k.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(dict)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if "k_pj" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:60], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for (k, i), v in sorted(d.items()):
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print("%-62s dispatch %-4s hit %14.6g miss %14.6g hit rate %.3f" % (k, i, h, m, h / (h + m) if h + m else 0))
PY
  find $O/pmc_tcc_k$k -name "*.csv" -size +1M -delete
done
echo "pmc done $(( $(date +%s) - t0 )) s" | tee -a $L
python - <<'PY' | tee -a $L
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_run1_bench_*.jsonl")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("r5_run1_bench_")[1], round(d["ms_per_step"], 3), "ms", {k[:20]: round(v, 2) for k, v in (r.get("kernels_ms") or {}).items()}, "build", d.get("join_build_ms"), "|", str(d.get("checked"))[:150])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 $O/r5_run1_tests_a.log $O/r5_run1_tests_b.log $O/r5_run1_cpp_api.log
grep -A17 "k_pj2_probe_pipe\|k_pj2_scatter" $O/r5_run1_pmc_sq_join_k0.txt | head -40
grep -A17 "k_pj3_probe_direct" $O/r5_run1_pmc_sq_join_k2.txt | head -40
cat $O/r5_run1_pmc_tcc_join_k0.txt $O/r5_run1_pmc_tcc_join_k2.txt
grep -E "Error|error|Traceback|assert" $L | head -20
