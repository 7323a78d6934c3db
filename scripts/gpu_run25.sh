#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run25.log
: > $L
cd /tmp && export TMPDIR=/tmp
for e in 0 1 2 4 6; do
GX_EXP=$e rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof25_$e -o join -- python $GRAFT_REPO_ROOT/bench.py --workload join --rows 1e9 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "EXP=$e" >> $GRAFT_REPO_ROOT/$L
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $GRAFT_REPO_ROOT/$O/prof25_$e/join_results.db 2>&1 | grep -E "k_pj_probe|k_pj_scatter" | cut -c1-50,105-170 >> $GRAFT_REPO_ROOT/$L
rm -rf $GRAFT_REPO_ROOT/$O/prof25_$e
done
cat $GRAFT_REPO_ROOT/$L
