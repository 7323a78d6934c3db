#!/bin/bash
# round 1, GPU call 3: tile-order experiments (sort), XCD-ranged groupby scatter A/B, PMC traffic
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
L=$O/run3.log
: > $L
B=$O/bench3.jsonl
: > $B
timeout 600 python -m pytest tests/test_gpu_join_groupby.py tests/test_gpu_reduce_scan_hash.py -m gpu -x -q > $O/pytest_gpu3.log 2>&1
echo "pytest exit $?" | tee -a $L
tail -5 $O/pytest_gpu3.log | tee -a $L
prof() { # name, args...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$name" -o $name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
  db=$(find $O/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 1 run 3: rocprofv3 --kernel-trace --stats -- python bench.py $*" > $O/r1_run3_${name}_kernel_stats.txt
  find $O/prof_$name -name "*.db" -delete
}
prof sort_a1_plain --rows 1e9 --steps 2 --warmup 1 --algo 17
prof sort_a1_ticket --rows 1e9 --steps 2 --warmup 1 --algo 33
prof groupby_ranged --workload groupby --rows 1e9 --steps 2 --warmup 1
prof groupby_single --workload groupby --rows 1e9 --steps 2 --warmup 1 --gb-algo 16
pmc() { # name, counter, args...
  local name=$1; shift
  local ctr=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d "$GRAFT_REPO_ROOT/$O/pmc_$name" -o $name --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline) >> $L 2>&1
}
pmc sort_a0_fetch FETCH_SIZE --rows 1e9 --steps 1 --warmup 1 --algo 0
pmc sort_a0_write WRITE_SIZE --rows 1e9 --steps 1 --warmup 1 --algo 0
pmc sort_a1_write WRITE_SIZE --rows 1e9 --steps 1 --warmup 1 --algo 1
pmc sort_a1_fetch FETCH_SIZE --rows 1e9 --steps 1 --warmup 1 --algo 1
ls -R $O/pmc_* | head -40 >> $L
for f in $(find $O/pmc_* -name "*counter_collection.csv"); do
  echo "== $f" >> $O/pmc3_summary.txt
  python - "$f" >> $O/pmc3_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = (r["Kernel_Name"][:90], r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items(), key=lambda x: -x[1][1])[:12]:
    print("%-92s %-12s calls=%4d  avg=%.6g" % (k, c, n, v / n))
PY
done
find $O/pmc_* -name "*.csv" -size +2M -delete
cat $O/pmc3_summary.txt
cat $O/r1_run3_*_kernel_stats.txt | grep -E "^#|k_radix_pass|k_tile_hist|k_part_" | cut -c1-170
tail -5 $L
