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
pmc p_sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" --workload join --rows 1e9 --steps 1 --warmup 0
pmc p_sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" --workload join --rows 1e9 --steps 1 --warmup 0
for f in $(find $O/pmc_p_* -name "*counter_collection.csv"); do
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pj" in r["Kernel_Name"]:
        print("%-34s %-22s %12.6g" % (r["Kernel_Name"][:34], r["Counter_Name"], float(r["Counter_Value"])))
PY
done > $O/pmcNN_summary.txt 2>&1
find $O/pmc_p_* -name "*.csv" -size +2M -delete
cat $O/pmcNN_summary.txt
tail -5 $L
