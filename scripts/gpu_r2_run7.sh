#!/bin/bash
# round 2 run 7: persistent gang probe (pieces in partition order per XCD), streaming reduce; join tests + bench + traffic
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run7.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_join_kinds_multikey.py tests/test_gpu_parity_1e8.py tests/test_gpu_reduce_scan_hash.py -m gpu -q -x -k "join or reduce or scan" > $O/pytest_gpu7.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu7.log | head -20 | tee -a $L
grep -E "^E  " $O/pytest_gpu7.log | head -20 | tee -a $L
: > $O/bench7.jsonl
timeout 300 python bench.py --workload join --no-cpu-baseline >> $O/bench7.jsonl 2>> $L
timeout 300 python bench.py --workload reduce --no-cpu-baseline >> $O/bench7.jsonl 2>> $L
python - <<'PY'
import json
for l in open('gpurun_out/bench7.jsonl'):
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '|', round(d['ms_per_step'], 2), 'ms | frac', round(r.get('frac', 0), 3), {k[:12]: round(v, 2) for k, v in r.get('kernels_ms', {}).items()})
PY
pmc() { local name=$1; shift; local ctr=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1; }
pmc j7_fetch FETCH_SIZE --workload join --rows 1e9 --steps 1 --warmup 0
pmc j7_tcc "TCC_HIT_sum TCC_MISS_sum" --workload join --rows 1e9 --steps 1 --warmup 0
for f in $(find $O/pmc_j7_* -name "*counter_collection.csv"); do
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pj_probe" in r["Kernel_Name"]:
        print("%-44s %-14s %12.6g" % (r["Kernel_Name"][:44], r["Counter_Name"], float(r["Counter_Value"])))
PY
done | tee $O/pmc7_join_probe.txt
find $O/pmc_j7_* -name "*.csv" -size +2M -delete
grep -E "exit|Error|error" $L | head
