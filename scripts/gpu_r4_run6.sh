#!/bin/bash
# round 4, run 6: the WINDOW build of the join (k_bw_split / k_bw_build) under its tests, the distributed loopback join (payload
# form of the build), and the join bench with the three build kernels (A/B on one box) + a kernel trace
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_join_build_subtable.py -x -q --durations=6 2>&1 | tail -16
timeout 600 python -m pytest tests/test_gpu_join_partition_modes.py tests/test_gpu_join_kinds_multikey.py -x -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "join and not 9_000_000" 2>&1 | tail -4
} > $O/r4_run6_tests.log 2>&1
for bk in 0 2 1; do
  python bench.py --workload join --no-cpu-baseline --steps 5 --join-build-kernel $bk > $O/r4_run6_bench_join_build$bk.jsonl 2> $O/r4_run6_err_$bk.txt
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_join6" -o join -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --workload join) > $O/r4_run6_join_under_rocprof.jsonl 2> $O/r4_run6_rocprof.err
db=$(find $O/prof_join6 -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 4 run 6: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload join (window build)" | grep -E "gx::|kernel|#|fillBuffer" | head -24 | cut -c1-200 > $O/r4_run6_join_kernel_stats.txt
rm -rf $O/prof_join6
cat $O/r4_run6_tests.log
python - <<'PY'
import json
for bk in (0, 2, 1):
    f = f"gpurun_out/r4_run6_bench_join_build{bk}.jsonl"
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("build kernel", bk, {k: round(d.get(k), 3) for k in ("ms_per_step", "join_build_ms", "join_build_call_ms", "join_build_plus_probe_ms")})
    except Exception as e:
        print(f, "unreadable", e, open(f"gpurun_out/r4_run6_err_{bk}.txt").read()[-500:])
PY
cut -c1-60,105-170 $O/r4_run6_join_kernel_stats.txt
