#!/bin/bash
# round 4, run 29: the PMC passes once more on the final kernels, now with the sorted_order workload (its block of the default line
# had no entry to attach): FETCH_SIZE / WRITE_SIZE in separate runs of sort / sorted_order / join / groupby at 1e9 rows
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
O=gpurun_out; L=$O/r4_run29.log; : > $L
pmc() { local wl=$1; local ctr=$2; local lc=$(echo $ctr | tr 'A-Z' 'a-z')
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_${wl}_${lc}" -o $wl --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --workload $wl --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline) >> $L 2>&1
}
for wl in sort sorted_order join groupby; do
  pmc $wl FETCH_SIZE
  pmc $wl WRITE_SIZE
done
python scripts/pmc_to_json.py $O $O/r4_pmc_traffic_1e9.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of python bench.py --workload <w> --rows 1e9 --steps 1 --warmup 0 (scripts/gpu_r4_run29.sh)" | tee $O/r4_run29_pmc_traffic.txt | grep GROUP
find $O/pmc_* -name "*.csv" -size +1M -delete
