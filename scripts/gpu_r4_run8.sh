#!/bin/bash
# round 4, run 8: state of the default bench line (sorted_order block included, CPU legs on), the C++ API suite (threaded loopback
# case), sorted_order on [0, 1e12) at 1e9 rows (DESIGN's unmeasured cliff), the contract / loopback tests after today's changes
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
O=gpurun_out
( time timeout 900 python bench.py ) > $O/r4_run8_bench_default.jsonl 2> $O/r4_run8_bench_default.err
timeout 300 python bench.py --workload sorted_order --no-cpu-baseline --key-range 0 1000000000000 > $O/r4_run8_bench_sorted_order_range1e12.jsonl 2>> $O/r4_run8_bench_default.err
timeout 600 tests/cpp/cudf_api_tests > $O/r4_run8_cpp_api_tests.log 2>&1; echo "cpp exit $?" >> $O/r4_run8_cpp_api_tests.log
timeout 600 python -m pytest tests/test_gpu_partition_reduce_contract.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -4 > $O/r4_run8_tests.log
tail -6 $O/r4_run8_cpp_api_tests.log; cat $O/r4_run8_tests.log; grep real $O/r4_run8_bench_default.err
python - <<'PY'
import json
for f in ("gpurun_out/r4_run8_bench_default.jsonl", "gpurun_out/r4_run8_bench_sorted_order_range1e12.jsonl"):
    for line in open(f):
        if not line.startswith("{"): continue
        d = json.loads(line)
        r = d.get("roofline") or {}
        print(d["config"]["workload"][:70], round(d["ms_per_step"], 3), "ms frac", round(r.get("frac", 0), 3), "path_frac", round(r.get("path_frac", 0), 3), (r.get("sort_info") or {}).get("cursor_path_state"), (r.get("sort_info") or {}).get("lsd_passes"))
        for k in ("sorted_order", "join", "groupby"):
            if k in d:
                rr = d[k].get("roofline") or {}
                print("  ", k, round(d[k]["ms_per_step"], 3), "ms frac", round(rr.get("frac", 0), 3), {kk: (round(v, 2) if isinstance(v, float) else v) for kk, v in d[k].items() if kk.startswith("build")})
        cb = d.get("cpu_baseline") or {}
        print("   cpu:", {k: (round(v) if isinstance(v, float) else v) for k, v in cb.items() if k != "sample"})
PY
