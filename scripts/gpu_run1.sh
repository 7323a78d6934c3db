#!/bin/bash
# First GPU contact: tiny sort first (bounded), then the gpu test-suite, then short benches + rocprof.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== device" | tee gpurun_out/run1.log
rocm-smi --showproductname 2>&1 | head -8 >> gpurun_out/run1.log
echo "== tiny sort (bounded)" | tee -a gpurun_out/run1.log
timeout 300 python - >> gpurun_out/run1.log 2>&1 <<'PY'
import numpy as np, torch, time
import cudf_amd
from cudf_amd import Column, ops, _lib
from oracle import cudf_oracle as orc
for algo in (1, 0):
    _lib.lib.gx_sort_set_algorithm(algo)
    for n in (1000, 8192, 100_000, 3_000_000):
        v = np.random.default_rng(1).integers(-2**63, 2**63-1, n, dtype=np.int64)
        t=time.time(); out = ops.sort(Column.from_numpy(v)).to_numpy(); dt=time.time()-t
        print("algo", algo, "n", n, "ok", out.tobytes()==np.sort(v).tobytes(), "%.3fs"%dt, flush=True)
PY
echo "exit $?" >> gpurun_out/run1.log
echo "== pytest -m gpu" | tee -a gpurun_out/run1.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/run1.log
tail -15 gpurun_out/pytest_gpu.log >> gpurun_out/run1.log
echo "== bench" | tee -a gpurun_out/run1.log
for algo in 0 1; do
  timeout 300 python bench.py --rows 1e8 --steps 3 --warmup 1 --algo $algo --no-cpu-baseline >> gpurun_out/bench_1e8.jsonl 2>> gpurun_out/run1.log
done
timeout 600 python bench.py --rows 1e9 --steps 3 --warmup 1 --algo 0 --no-cpu-baseline >> gpurun_out/bench_1e9.jsonl 2>> gpurun_out/run1.log
timeout 600 python bench.py --rows 1e9 --steps 3 --warmup 1 --algo 1 --no-cpu-baseline >> gpurun_out/bench_1e9.jsonl 2>> gpurun_out/run1.log
timeout 300 python bench.py --workload sorted_order --rows 1e8 --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/bench_1e8.jsonl 2>> gpurun_out/run1.log
timeout 300 python bench.py --workload join --rows 1e8 --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/bench_1e8.jsonl 2>> gpurun_out/run1.log
timeout 300 python bench.py --workload groupby --rows 1e8 --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/bench_1e8.jsonl 2>> gpurun_out/run1.log
echo "== rocprof" | tee -a gpurun_out/run1.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_sort" -o sort1e9 -- python "$OLDPWD/bench.py" --rows 1e9 --steps 2 --warmup 1 --no-cpu-baseline) >> gpurun_out/run1.log 2>&1
ls -R gpurun_out | head -40 >> gpurun_out/run1.log
cat gpurun_out/bench_1e8.jsonl gpurun_out/bench_1e9.jsonl
tail -40 gpurun_out/run1.log
