#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run11.log
: > $L
timeout 900 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_distributed.py tests/test_cpp_api.py -m gpu -x -q > $O/pytest_gpu11.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -6 $O/pytest_gpu11.log | tee -a $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 11: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run11_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof join_part --workload join --rows 1e9 --steps 2 --warmup 1
python bench.py --workload join --rows 1e9 --steps 2 --warmup 1 --no-cpu-baseline --no-partitioned-join >> $L 2>&1
cat $O/r1_run11_*_kernel_stats.txt | grep -E "^# round|k_pj|k_probe|k_build" | cut -c1-170
grep -h '"metric"' $L | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline'] or {}
    print(d['config']['workload'][:60], '| ms', round(d['ms_per_step'],2), '| Grows/s', round(d['value']/1e9,2), '| frac', round(r.get('frac',0),3), '| bits', d.get('join_partition_bits'))
"
