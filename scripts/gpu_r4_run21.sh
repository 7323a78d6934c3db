#!/bin/bash
# round 4, run 21: the whole GPU suite (python -m pytest tests -m gpu, as the driver runs it) + the C++ API tests, on the final kernels
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -40 ) > $O/r4_run21_full_gpu_suite.log 2>&1
( timeout 600 tests/cpp/cudf_api_tests 2>&1 | grep -v amdgpu.ids | tail -8; echo "cpp exit ${PIPESTATUS[0]}" ) > $O/r4_run21_cpp_api_tests.log 2>&1
tail -12 $O/r4_run21_full_gpu_suite.log
tail -5 $O/r4_run21_cpp_api_tests.log
