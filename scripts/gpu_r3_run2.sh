#!/bin/bash
# round 3 run 2: the probe's per-wave deferral queue (rows not settled by their first candidate slot wait one trip instead
# of stalling their wave): tests of every knob combination incl. full queues, then the join A/B on random and dense keys
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_join_partition_modes.py -q --durations=5 > $O/r3_run2_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -15 $O/r3_run2_pytest.log | tee -a $L
: > $O/r3_run2_bench_join_ab.jsonl
for cfg in "1 1 random" "1 0 random" "1 3 random" "1 1 dense" "1 0 dense"; do
  set -- $cfg
  echo "== join spec=$1 mode=$2 keys=$3" | tee -a $L
  timeout 600 python bench.py --workload join --no-cpu-baseline --join-spec $1 --join-early-loads $2 --join-keys $3 >> $O/r3_run2_bench_join_ab.jsonl 2>> $L
done
python - <<'PY' | tee -a gpurun_out/r3_run2.log
import json
for line in open('gpurun_out/r3_run2_bench_join_ab.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    r = d.get('roofline') or {}
    print(d.get('join_keys'), d.get('join_partition_mode'), round(d['ms_per_step'], 3), 'ms', {k[:22]: round(v, 3) for k, v in (r.get('kernels_ms') or {}).items()}, 'build', round(d.get('join_build_ms', 0), 2))
PY
