#!/bin/bash
# SQ counters (issue / wait / LDS) of the kernels of one bench workload, two --pmc passes (8 SQ slots each).
# usage (on the GPU box, via gpurun):  bash scripts/gpu_pmc_sq.sh <workload> <kernel-name-substring> [bench args...]
#   e.g.  bash scripts/gpu_pmc_sq.sh sort gx::sort        bash scripts/gpu_pmc_sq.sh join k_pj
# --pmc is never combined with sys / hip / hsa traces (only --kernel-trace).
set -u
W=${1:-sort}; PAT=${2:-gx::}; shift 2 || true
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/pmc_sq_$W.log
: > $L
pmc() { # name, counters
  local name=$1; shift
  local ctr=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload $W --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline "$@") >> $L 2>&1
}
pmc sq1_$W "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "$@"
pmc sq2_$W "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "$@"
python - "$PAT" $(find $O/pmc_sq1_$W $O/pmc_sq2_$W -name "*counter_collection.csv") > $O/pmc_sq_${W}_summary.txt <<'PY'
import collections, csv, sys
pat, files = sys.argv[1], sys.argv[2:]
d = collections.defaultdict(dict)
for f in files:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:70], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for (k, i), v in sorted(d.items()):
    print(k, "dispatch", i)
    for a, b in sorted(v.items()):
        print("    %-22s %14.6g" % (a, b))
PY
find $O/pmc_sq1_$W $O/pmc_sq2_$W -name "*.csv" -size +2M -delete
head -120 $O/pmc_sq_${W}_summary.txt
tail -3 $L
