#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run24.log
: > $L
timeout 900 python -m pytest tests/test_gpu_join_groupby.py -m gpu -x -q -k "join or probe" > $O/pytest_gpu24.log 2>&1
echo "pytest join exit $?" | tee -a $L
tail -12 $O/pytest_gpu24.log | tee -a $L
for sc in 8 1 4 16; do
GX_PJ_SC=$sc python bench.py --workload join --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline >> $L 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof24 -o join -- python $GRAFT_REPO_ROOT/bench.py --workload join --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_summary.py $O/prof24/join_results.db > $O/prof24_summary.txt 2>&1
head -12 $O/prof24_summary.txt | cut -c1-60,105-170 | tee -a $L
grep -h '"metric"' $L | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['config']['workload'][:40], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '|', r['kernel'][:30], round(r['avg_launch_ms'],2), d.get('join_build_ms'))
"
