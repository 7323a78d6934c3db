#!/bin/bash
# round 3 run 24: 16-bit per-wave counters in the stable partition passes (the pairs' 9-bit level 1: 82 000 -> 73 808 B of LDS,
# two workgroups per CU), one-barrier scan + verdict in k_local_place -- the FULL GPU suite, then the A/B at 1e9 rows
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run24.log
: > $L
timeout 200 python scripts/xp/xp_place_ab.py 1e9 both i64 2>&1 | grep -v amdgpu.ids | tee $O/r3_run24_place_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > $O/r3_run24_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -18 $O/r3_run24_pytest.log | tee -a $L
