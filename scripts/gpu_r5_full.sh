#!/bin/bash
# round 5: the full GPU suite + the default bench line as the driver runs it (wall clock)
set -u
R=${1:-10}
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
L=$O/r5_run${R}.log
: > $L
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $O/r5_run${R}_full_gpu_suite.log 2>&1
echo "pytest exit $?" | tee -a $L
grep -E "^(FAILED|ERROR)|passed|failed|^real" $O/r5_run${R}_full_gpu_suite.log | head -30 | tee -a $L
( time timeout 900 python bench.py ) > $O/r5_run${R}_bench_default.jsonl 2>> $L
grep -E "^real" $L | tail -1
python - $R <<'PY' | tee -a $L
import json, sys
R = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r5_run{R}_bench_default.jsonl") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("sort", round(d["ms_per_step"], 3), "ms frac", round(r.get("frac", 0), 3), {k[:22]: round(v, 2) for k, v in (r.get("kernels_ms") or {}).items()})
for k in ("sorted_order", "join", "groupby"):
    if k in d:
        print(" ", k, round(d[k]["ms_per_step"], 3), "ms frac", round((d[k].get("roofline") or {}).get("frac", 0), 3), str(d[k].get("checked"))[:100])
print("  through_cpp", d.get("through_cpp"))
rb = d.get("sort_robustness") or {}
print("  robustness worst", rb.get("worst"))
for k, v in (rb.get("cases") or {}).items():
    print("     ", k, v)
PY
grep -E "Error|error|Traceback" $L | head
