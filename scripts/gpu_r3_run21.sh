#!/bin/bash
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python scripts/xp/xp_local_sort_ablation.py 1e9 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_run21_local_sort_ablation.txt
