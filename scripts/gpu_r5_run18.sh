#!/bin/bash
# round 5 run 18: the look-back guard as a recoverable status (fault injection), the C++ API tests, the drop-in surface with the status read-back
set -u
R=${1:-18}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
( timeout 900 python -m pytest tests/test_gpu_sort_fault.py tests/test_cpp_api.py tests/test_gpu_sort.py -m gpu -q -x 2>&1 | tail -30 ) > $O/r5_run${R}_tests.log
tail -n 12 $O/r5_run${R}_tests.log
timeout 600 python bench.py --workload sort --no-cpu-baseline --no-robustness --through-cpp 2>> $L | tail -1 > $O/r5_run${R}_bench_sort_through_cpp.jsonl
timeout 600 python bench.py --workload sorted_order --no-cpu-baseline --no-robustness --through-cpp 2>> $L | tail -1 > $O/r5_run${R}_bench_sorted_order_through_cpp.jsonl
python - $R <<'PY'
import json, sys
R = sys.argv[1]
for w in ("sort", "sorted_order"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r5_run{R}_bench_{w}_through_cpp.jsonl") if l.startswith("{")][-1])
        print(w, round(d["ms_per_step"], 3), "ms | through_cpp:", json.dumps(d.get("through_cpp"))[:600])
    except Exception as e:
        print(w, "unreadable", e)
PY
grep -E "Error|error|Traceback|assert" $L | head -20
