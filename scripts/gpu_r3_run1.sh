#!/bin/bash
# round 3 run 1: the new parity tests (C++ surface through the shim; speculative join partition + fallback) and the
# join A/B on SURVEY 8d's random keys: speculative/early-loads vs the round-2 path, dense keys beside them
set -u
ulimit -c 0   # a crashing run must not fill the box's disk with core files
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run1.log
: > $L
timeout 900 python -m pytest tests/test_gpu_join_partition_modes.py tests/test_gpu_cpp_parity.py -q --durations=8 > $O/r3_run1_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -25 $O/r3_run1_pytest.log | tee -a $L
: > $O/r3_run1_bench_join_ab.jsonl
for cfg in "1 1 random" "1 0 random" "0 1 random" "0 1 dense" "1 1 dense"; do
  set -- $cfg
  echo "== join spec=$1 early=$2 keys=$3" | tee -a $L
  timeout 600 python bench.py --workload join --no-cpu-baseline --join-spec $1 --join-early-loads $2 --join-keys $3 >> $O/r3_run1_bench_join_ab.jsonl 2>> $L
done
python - <<'PY' | tee -a gpurun_out/r3_run1.log
import json
for line in open('gpurun_out/r3_run1_bench_join_ab.jsonl'):
    try: d = json.loads(line)
    except Exception: continue
    r = d.get('roofline') or {}
    print(d.get('join_keys'), d.get('join_partition_mode'), round(d['ms_per_step'], 3), 'ms', {k[:22]: round(v, 3) for k, v in (r.get('kernels_ms') or {}).items()}, 'build', round(d.get('join_build_ms', 0), 2))
PY
