#!/bin/bash
# round 4, run 2: the sub-table join build (k_bs_build) under its tests, the contract tests of hash_partition / reduce(init),
# the 1e8-row join parity on random keys, and the join bench with both build kernels (A/B on one box)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_join_build_subtable.py tests/test_gpu_partition_reduce_contract.py -x -q --durations=8 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_parity_1e8.py -x -q -k join 2>&1 | tail -5
} > gpurun_out/r4_run2_tests.log 2>&1
python bench.py --workload join --no-cpu-baseline --steps 5 > gpurun_out/r4_run2_bench_join_build_subtable.jsonl 2> gpurun_out/r4_run2_bench0.err
python bench.py --workload join --no-cpu-baseline --steps 5 --join-build-kernel 1 > gpurun_out/r4_run2_bench_join_build_round2.jsonl 2> gpurun_out/r4_run2_bench1.err
tail -30 gpurun_out/r4_run2_tests.log
python - <<'PY'
import json
for f in ("gpurun_out/r4_run2_bench_join_build_subtable.jsonl", "gpurun_out/r4_run2_bench_join_build_round2.jsonl"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("ms_per_step", "join_build_ms", "join_build_call_ms", "join_build_plus_probe_ms")})
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r4_run2_bench0.err
