#!/bin/bash
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
GXD_TRACE=1 timeout 600 python scripts/xp/xp_gxd_alloc_probe.py 1e9 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tee gpurun_out/r3_run14_alloc_probe.txt
