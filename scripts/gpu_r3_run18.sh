#!/bin/bash
# round 3 run 18/19: wide groupby (claim/release restructured, sampled slot capacities): raw verdicts + tests under per-test timeouts
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=$O/r3_run18.log
: > $L
timeout 120 python scripts/xp/xp_groupby_wide_probe.py 1e8 1e9 2>&1 | grep -v amdgpu.ids | tee -a $L
timeout 600 python -m pytest tests/test_gpu_groupby_wide.py tests/test_gpu_dataframe.py -q -x --timeout 90 > $O/r3_run18_pytest.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -30 $O/r3_run18_pytest.log | cut -c1-200 | tee -a $L
echo finished | tee -a $L
