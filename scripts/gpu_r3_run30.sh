#!/bin/bash
# round 3 run 30: sorted_order of 32-bit integer keys as a 64-bit word sort -- parity, then 1e9 rows against the LSD pair passes
# (unique-ish keys and 1e6 distinct keys)
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run30.log
: > $L
timeout 600 python -m pytest tests/test_gpu_sort_place.py -m gpu -q -x -k "word_sort or 32bit" > $O/r3_run30_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -8 $O/r3_run30_pytest.log | tee -a $L
timeout 200 python scripts/xp/xp_order32.py 1e9 2>&1 | grep -v amdgpu.ids | tee $O/r3_run30_order32.txt
timeout 200 python scripts/xp/xp_order32.py 1e9 1e6 2>&1 | grep -v amdgpu.ids | tee -a $O/r3_run30_order32.txt
