#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run8.log
: > $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 8: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run8_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
GX_EXP=1 prof exp1 --rows 1e9 --steps 2 --warmup 1
GX_EXP=2 prof exp2 --rows 1e9 --steps 2 --warmup 1
pmc() { # name, counters, args...
  local name=$1; shift
  local ctr=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
}
pmc hy_fetch FETCH_SIZE --rows 1e9 --steps 1 --warmup 0
pmc hy_write WRITE_SIZE --rows 1e9 --steps 1 --warmup 0
pmc hy_tcc "TCC_HIT_sum TCC_MISS_sum" --rows 1e9 --steps 1 --warmup 0
for f in $(find $O/pmc_hy_* -name "*counter_collection.csv"); do
  echo "== $f"
  python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_msd_pass" in r["Kernel_Name"] or "k_local_sort" in r["Kernel_Name"] or "k_hist2" in r["Kernel_Name"]]
for r in rows:
    print("%-40s %-14s %12.6g  dispatch %s" % (r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"]), r.get("Dispatch_Id")))
PY
done > $O/pmc8_summary.txt 2>&1
find $O/pmc_hy_* -name "*.csv" -size +2M -delete
cat $O/pmc8_summary.txt
grep -h "kernels_ms" $L | sed 's/.*"kernels_ms": //; s/, "kernels_GBps".*//'
tail -3 $L
