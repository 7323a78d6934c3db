#!/bin/bash
# round 3 run 6: (a) kernel trace of the chunked gxd_sort (why 253 ms with >= 4 chunks?), (b) the full GPU test suite on the
# current tree (incl. sorted_order on 8192-key cells / 9-bit level 1 for pairs), (c) sorted_order + default bench lines
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run6.log
: > $L
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_gxdsort" -o gxdsort -- python "$GRAFT_REPO_ROOT/scripts/xp/xp_gxd_sort_trace.py" 1e9 4) >> $L 2>&1
db=$(find $O/prof_gxdsort -name "*.db" | head -1)
if [ -n "$db" ]; then
  python scripts/rocprof_summary.py "$db" "round 3 run 6: rocprofv3 --kernel-trace --stats -- xp_gxd_sort_trace.py 1e9 4" | head -30 | cut -c1-200 > $O/r3_run6_gxd_sort_chunked_kernel_stats.txt
  cat $O/r3_run6_gxd_sort_chunked_kernel_stats.txt | tee -a $L
fi
find $O/prof_gxdsort -name "*.db" -delete
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/r3_run6_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -22 $O/r3_run6_pytest.log | tee -a $L
timeout 600 python bench.py --workload sorted_order --no-cpu-baseline > $O/r3_run6_bench_sorted_order.jsonl 2>> $L
timeout 900 python bench.py --no-cpu-baseline > $O/r3_run6_bench_default.jsonl 2>> $L
python - <<'PY' | tee -a gpurun_out/r3_run6.log
import json
for f in ('gpurun_out/r3_run6_bench_sorted_order.jsonl', 'gpurun_out/r3_run6_bench_default.jsonl'):
    for line in open(f):
        try: d = json.loads(line)
        except Exception: continue
        print(d['config']['workload'][:60], round(d['ms_per_step'], 3), 'ms', {k[:18]: round(v, 2) for k, v in ((d.get('roofline') or {}).get('kernels_ms') or {}).items()})
        for k in ('join', 'groupby'):
            if k in d: print('  ', k, round(d[k]['ms_per_step'], 3), 'ms', {kk[:18]: round(v, 2) for kk, v in ((d[k].get('roofline') or {}).get('kernels_ms') or {}).items()})
PY
