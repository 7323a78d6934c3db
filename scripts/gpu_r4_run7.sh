#!/bin/bash
# round 4, run 7: the fused sharded sort (gx_sortx_* + gxd_sort) under the loopback tests, then the forced single-rank steps
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_distributed_loopback.py -x -q -k "sort" --durations=5 > $O/r4_run7_tests.log 2>&1
tail -25 $O/r4_run7_tests.log
timeout 600 python scripts/xp/xp_gxd_steps.py > $O/r4_run7_single_rank_steps.txt 2>&1
cat $O/r4_run7_single_rank_steps.txt | grep -v "^\[W\|amdgpu.ids" | tail -40
