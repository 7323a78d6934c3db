#!/bin/bash
# round 4, run 28: last check of the committed tree as the driver will use it: smoke(), then the default bench line (traffic must be
# attached from the committed PMC file: same kernel sources)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r4_run28_bench_default.jsonl
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_run28_bench_default.jsonl").read())
r = d["roofline"]
print(round(d["ms_per_step"], 3), "ms frac", round(r["frac"], 3), "traffic", r.get("traffic"), r.get("traffic_note"), (r.get("traffic_source") or "")[:60])
for k in ("sorted_order", "join", "groupby"):
    rr = d[k]["roofline"]
    print(k, round(d[k]["ms_per_step"], 3), "frac", round(rr["frac"], 3), "traffic", rr.get("traffic"), rr.get("traffic_note"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
