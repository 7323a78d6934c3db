#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/runNN.log
: > $L
pmc() { # name, counters, args...
  local name=$1; shift
  local ctr=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
}
pmc j_tcc "TCC_HIT_sum TCC_MISS_sum" --workload join --rows 1e9 --steps 1 --warmup 0
pmc j_fetch FETCH_SIZE --workload join --rows 1e9 --steps 1 --warmup 0
pmc j_write WRITE_SIZE --workload join --rows 1e9 --steps 1 --warmup 0
for f in $(find $O/pmc_j_* -name "*counter_collection.csv"); do
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pj" in r["Kernel_Name"] or "k_build" in r["Kernel_Name"]:
        print("%-34s %-14s %12.6g" % (r["Kernel_Name"][:34], r["Counter_Name"], float(r["Counter_Value"])))
PY
done > $O/pmcNN_summary.txt 2>&1
find $O/pmc_j_* -name "*.csv" -size +2M -delete
cat $O/pmcNN_summary.txt
