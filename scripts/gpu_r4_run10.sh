#!/bin/bash
# round 4, run 10: groupby partition bits chosen on the device (tests + dense / sparse lines in the default mode), loopback join tests
# with the C oracle
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_groupby_wide.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "join or world" --durations=4 2>&1 | tail -9
timeout 600 python -m pytest tests/test_gpu_parity_1e8.py tests/test_gpu_dataframe.py -x -q -k "groupby or dataframe" 2>&1 | tail -3
} > $O/r4_run10_tests.log 2>&1
for keys in dense random random64; do
  python bench.py --workload groupby --no-cpu-baseline --steps 5 --gb-keys $keys > $O/r4_run10_bench_groupby_auto_$keys.jsonl 2> $O/r4_run10_err.txt
done
python bench.py --workload groupby --no-cpu-baseline --steps 5 --gb-pbits 9 > $O/r4_run10_bench_groupby_pbits9_dense.jsonl 2>> $O/r4_run10_err.txt
cat $O/r4_run10_tests.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_run10_bench_*.jsonl")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("r4_run10_bench_")[1], round(d["ms_per_step"], 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $O/r4_run10_err.txt
