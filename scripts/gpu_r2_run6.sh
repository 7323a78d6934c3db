#!/bin/bash
# round 2 run 6: C++ API cases (partitioned joins, sort-path groupby fix), join traffic (FETCH / WRITE) + SQ counters
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run6.log
: > $L
timeout 600 python -m pytest tests/test_cpp_api.py -m gpu -q > $O/pytest_gpu6.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "passed|failed|CHECK failed|FAIL\]|what\(\)" $O/pytest_gpu6.log | head -20 | tee -a $L
pmc() { # name, counters, args...
  local name=$1; shift
  local ctr=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
}
pmc j_fetch FETCH_SIZE --workload join --rows 1e9 --steps 1 --warmup 0
pmc j_write WRITE_SIZE --workload join --rows 1e9 --steps 1 --warmup 0
pmc j_tcc "TCC_HIT_sum TCC_MISS_sum" --workload join --rows 1e9 --steps 1 --warmup 0
for f in $(find $O/pmc_j_* -name "*counter_collection.csv"); do
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pj" in r["Kernel_Name"]:
        print("%-44s %-14s %12.6g" % (r["Kernel_Name"][:44], r["Counter_Name"], float(r["Counter_Value"])))
PY
done > $O/pmc6_join_traffic.txt 2>&1
find $O/pmc_j_* -name "*.csv" -size +2M -delete
cat $O/pmc6_join_traffic.txt
bash scripts/gpu_pmc_sq.sh join k_pj > /dev/null 2>&1
grep -A17 "k_pj_probe_pipe\|k_pj_scatter" $O/pmc_sq_join_summary.txt | head -80
