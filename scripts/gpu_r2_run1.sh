#!/bin/bash
# round 2 run 1: micro-benchmark of random tag reads (Infinity Cache), join tests on the new scatter / probe
# kernels, join A/B (probe kernel x scatter tile), then the default bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run1.log
: > $L
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/xp/xp_randread.hip -o /tmp/xp_randread >> $L 2>&1
timeout 120 /tmp/xp_randread 1e9 22 24 26 27 28 29 30 32 > $O/xp_randread.jsonl 2>> $L
cat $O/xp_randread.jsonl
timeout 900 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_join_kinds_multikey.py -m gpu -x -q > $O/pytest_gpu1.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/pytest_gpu1.log | tee -a $L
: > $O/bench1_join_ab.jsonl
for pk in 0 1; do for tile in 0 8192 4096; do
  timeout 300 python bench.py --workload join --no-cpu-baseline --join-probe-kernel $pk --join-scatter-tile $tile >> $O/bench1_join_ab.jsonl 2>> $L
  echo "join pk=$pk tile=$tile exit $?" >> $L
done; done
python - <<'PY'
import json
for l in open('gpurun_out/bench1_join_ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(round(d['ms_per_step'], 2), 'ms | frac', round(r['frac'], 3), '|', {k[:12]: round(v, 2) for k, v in r.get('kernels_ms', {}).items()}, '| build', round(d.get('join_build_ms', 0), 1))
PY
( time timeout 900 python bench.py ) > $O/bench1_default.jsonl 2>> $L
cut -c1-2500 $O/bench1_default.jsonl
grep -E "real|exit|Error|error" $L | head -20
