#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out
L=$O/run27.log
: > $L
cd /tmp && export TMPDIR=/tmp
for cfg in "128 8" "384 8"; do
set -- $cfg
GX_EXP=$1 GX_PJ_SC=$2 timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof27 -o join -- python $GRAFT_REPO_ROOT/bench.py --workload join --rows 1e9 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "EXP=$1 SC=$2" >> $GRAFT_REPO_ROOT/$L
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $GRAFT_REPO_ROOT/$O/prof27/join_results.db 2>&1 | grep -E "k_pj_probe|k_pj_scatter" | cut -c1-50,105-170 >> $GRAFT_REPO_ROOT/$L
rm -rf $GRAFT_REPO_ROOT/$O/prof27
done
cat $GRAFT_REPO_ROOT/$L
