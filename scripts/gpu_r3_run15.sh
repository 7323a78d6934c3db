#!/bin/bash
# round 3 run 15: forced single-rank steps after the result-buffer cycle fix (gxd.py) and the pooled join buffers
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python scripts/xp/xp_distributed_single_rank.py 1e9 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 | tee $O/r3_run15_single_rank_steps.txt
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -x 2>&1 | tail -3
