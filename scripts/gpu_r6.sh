#!/bin/bash
# Round 6: ONE parametrised GPU script (VERDICT r5 housekeeping: no more one-shot gpu_r*_run*.sh files).
#   usage: bash scripts/gpu_r6.sh <run tag> <step> [<step> ...]      -- every step writes gpurun_out/r6_<tag>_<step>.*
# steps
#   test:<pytest -k expr or file>   pytest -m gpu on tests/ (or one file) with -k <expr>
#   suite                           the whole -m gpu suite
#   cpp                             tests/cpp/cudf_api_tests through tests/test_cpp_api.py
#   joinab:<xp,xp,...>              bench.py --workload join for each --join-xp value (A/B on one box)
#   sortab:<dist,dist,...>          bench.py --workload sort --key-dist <d> on the tree's library, then on scripts/xp/bin/libcudf_amd_oldsort.so, then the tree's again
#   bench:<workload>[:extra args]   bench.py --workload <w> --no-cpu-baseline <extra>
#   default                         the driver-style default line (all legs)
#   steps:<rows>                    scripts/xp/xp_gxd_steps.py <rows>  (forced-exchange single-rank steps of the sharded operators)
#   profsteps:<rows>                rocprofv3 --kernel-trace --stats summary of scripts/xp/xp_gxd_steps.py <rows>
#   xp:<file.hip>[:args]            hipcc + run a micro-benchmark under scripts/xp/
#   overlap[:log2 rows]             rocprofv3 --kernel-trace of scripts/xp/bin/xp_two_streams (two sorts on two streams) + scripts/overlap_summary.py
#   prof:<workload>[:extra args]    rocprofv3 --kernel-trace --stats summary of bench.py --workload <w>
#   pmcsq:<workload>[:extra args]   SQ counters (two passes) of bench.py --workload <w> --steps 1 --warmup 0
#   evidence                        default line + sorted_order line + rocprof summary + FETCH/WRITE passes -> r6_pmc_traffic_1e9.json
set -u
ulimit -c 0
export HSA_ENABLE_COREDUMP=0 AMD_LOG_LEVEL=0 TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
O=gpurun_out
mkdir -p $O
TAG=${1:-x}
shift
last_json() { grep '^{' "$1" | tail -1; }
for step in "$@"; do
  kind=${step%%:*}
  rest=${step#*:}
  [ "$rest" = "$step" ] && rest=""
  case $kind in
    test)
      f=$O/r6_${TAG}_test.log
      if [ -f "$rest" ]; then timeout 1500 python -m pytest "$rest" -m gpu -q -x 2>&1 | tail -15 >> $f
      else timeout 1500 python -m pytest tests -m gpu -q -x -k "$rest" 2>&1 | tail -15 >> $f; fi
      tail -6 $f ;;
    suite)
      timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r6_${TAG}_full_gpu_suite.log
      tail -5 $O/r6_${TAG}_full_gpu_suite.log ;;
    cpp)
      timeout 900 python -m pytest tests/test_cpp_api.py -m gpu -q -x 2>&1 | tail -8 > $O/r6_${TAG}_cpp_api.log
      tail -3 $O/r6_${TAG}_cpp_api.log ;;
    joinab)
      out=$O/r6_${TAG}_join_ab.txt
      echo "# round 6 $TAG: python bench.py --workload join --no-cpu-baseline --join-xp <xp> (1e9 probe x 1e8 build rows, random 64-bit keys); ms per step | scatter | probe" >> $out
      for xp in ${rest//,/ }; do
        early=""; case $xp in *e) early="--join-early-loads 1"; xp=${xp%e};; *d) early="--join-early-loads 2"; xp=${xp%d};; *b) early="--join-early-loads 3"; xp=${xp%b};; esac
        unchk=""; [ $(( (xp >> 4) & 7 )) -ne 0 ] && unchk="--join-unchecked"
        [ -n "$early" ] && echo -n "$early " | tee -a $out
        timeout 400 python bench.py --workload join --no-cpu-baseline --join-xp $xp $unchk $early > $O/r6_${TAG}_bench_join_xp$xp.jsonl 2>> $O/r6_${TAG}.log
        python - "$O/r6_${TAG}_bench_join_xp$xp.jsonl" $xp <<'PY' | tee -a $out
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("xp", sys.argv[2], "|", round(d["ms_per_step"], 3), "ms |", {k[:24]: round(v, 3) for k, v in (r.get("kernels_ms") or {}).items()}, "| build", round(d.get("build_ms", 0), 2))
except Exception as e:
    print("xp", sys.argv[2], "| no line:", e)
PY
      done ;;
    sortab)
      # same-box A/B of two builds of the kernel library: the tree's, then scripts/xp/bin/libcudf_amd_oldsort.so (a build with another gx_sort.hip)
      out=$O/r6_${TAG}_sort_ab.txt
      echo "# round 6 $TAG: python bench.py --workload sort --no-cpu-baseline --no-robustness --no-through-cpp --key-dist <d>; ms per step, 1e9 int64 keys" >> $out
      for variant in new old new2; do
        [ $variant = old ] && { cp cudf_amd/libcudf_amd.so /tmp/libcudf_amd_new.so; cp scripts/xp/bin/libcudf_amd_oldsort.so cudf_amd/libcudf_amd.so; }
        [ $variant = new2 ] && cp /tmp/libcudf_amd_new.so cudf_amd/libcudf_amd.so
        for d in ${rest//,/ }; do
          timeout 300 python bench.py --workload sort --no-cpu-baseline --no-robustness --no-through-cpp --key-dist $d > $O/r6_${TAG}_sortab_${variant}_$d.jsonl 2>> $O/r6_${TAG}.log
          python - "$O/r6_${TAG}_sortab_${variant}_$d.jsonl" $variant $d <<'PY' | tee -a $out
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print(sys.argv[2], sys.argv[3], "|", round(d["ms_per_step"], 3), "ms |", {k[:22]: round(v, 3) for k, v in (r.get("kernels_ms") or {}).items()})
except Exception as e:
    print(sys.argv[2], sys.argv[3], "| no line:", e)
PY
        done
      done ;;
    bench)
      wl=${rest%%:*}; extra=${rest#*:}; [ "$extra" = "$rest" ] && extra=""
      f=$O/r6_${TAG}_bench_${wl}$(echo "$extra" | tr -c 'A-Za-z0-9\n' '_').jsonl
      timeout 600 python bench.py --workload $wl --no-cpu-baseline $extra > $f 2>> $O/r6_${TAG}.log
      last_json $f | cut -c1-600 ;;
    default)
      ( time timeout 900 python bench.py ) > $O/r6_${TAG}_bench_default.jsonl 2>> $O/r6_${TAG}.log
      last_json $O/r6_${TAG}_bench_default.jsonl | cut -c1-1500 ;;
    steps)
      timeout 900 python scripts/xp/xp_gxd_steps.py ${rest:-1e9} 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/r6_${TAG}_single_rank_steps_${rest:-1e9}.txt
      cat $O/r6_${TAG}_single_rank_steps_${rest:-1e9}.txt | tail -30 ;;
    profsteps)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof_$TAG" -o p -- python "$ROOT/scripts/xp/xp_gxd_steps.py" ${rest:-1e9}) > $O/r6_${TAG}_steps_under_rocprof.txt 2>> $O/r6_${TAG}.log
      db=$(find $O/prof_$TAG -name "*.db" | head -1)
      [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 6 $TAG: rocprofv3 --kernel-trace --stats -- python scripts/xp/xp_gxd_steps.py ${rest:-1e9}" | head -70 | cut -c1-190 > $O/r6_${TAG}_steps_kernel_stats.txt
      find $O/prof_$TAG -name "*.db" -delete
      head -50 $O/r6_${TAG}_steps_kernel_stats.txt ;;
    xp)
      src=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
      mkdir -p scripts/xp/bin
      b=scripts/xp/bin/$(basename $src .hip)
      [ -x $b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $b scripts/xp/$src
      timeout 600 $b $args 2>&1 | tee $O/r6_${TAG}_$(basename $src .hip).txt | tail -40 ;;
    overlap)
      # two cudf::sort calls on two streams under rocprofv3 --kernel-trace: do the kernels of the two streams run at the same time?
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$ROOT/$O/prof_$TAG" -o p -- "$ROOT/scripts/xp/bin/xp_two_streams" ${rest:-27}) > $O/r6_${TAG}_two_streams.txt 2>> $O/r6_${TAG}.log
      db=$(find $O/prof_$TAG -name "*.db" | head -1)
      [ -n "$db" ] && python scripts/overlap_summary.py "$db" "round 6 $TAG: rocprofv3 --kernel-trace -- scripts/xp/bin/xp_two_streams ${rest:-27} (two cudf::sort calls on two streams, 3 repetitions + one sort alone each)" | head -150 | cut -c1-170 >> $O/r6_${TAG}_two_streams.txt
      find $O/prof_$TAG -name "*.db" -delete
      head -60 $O/r6_${TAG}_two_streams.txt ;;
    prof)
      wl=${rest%%:*}; extra=${rest#*:}; [ "$extra" = "$rest" ] && extra=""
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof_$TAG" -o p -- python "$ROOT/bench.py" --workload $wl --no-cpu-baseline --no-robustness --no-through-cpp $extra) > $O/r6_${TAG}_bench_under_rocprof.jsonl 2>> $O/r6_${TAG}.log
      db=$(find $O/prof_$TAG -name "*.db" | head -1)
      [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 6 $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --no-cpu-baseline --no-robustness --no-through-cpp $extra" | head -60 | cut -c1-190 > $O/r6_${TAG}_${wl}_kernel_stats.txt
      find $O/prof_$TAG -name "*.db" -delete
      head -30 $O/r6_${TAG}_${wl}_kernel_stats.txt ;;
    pmcsq)
      wl=${rest%%:*}; extra=${rest#*:}; [ "$extra" = "$rest" ] && extra=""
      pat=gx::; [ "$wl" = join ] && pat=k_pj
      bash scripts/gpu_pmc_sq.sh $wl $pat $extra > /dev/null 2>> $O/r6_${TAG}.log
      ( echo "# round 6 $TAG: rocprofv3 --kernel-trace --pmc <SQ counters, two passes> -- python bench.py --workload $wl --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline $extra"; cat $O/pmc_sq_${wl}_summary.txt ) > $O/r6_${TAG}_pmc_sq_${wl}.txt
      grep -A17 "scatter\|probe_pipe\|k_hf_scatter\|k_local_place" $O/r6_${TAG}_pmc_sq_${wl}.txt | head -80 ;;
    evidence)
      ( time timeout 900 python bench.py ) > $O/r6_${TAG}_bench_default.jsonl 2>> $O/r6_${TAG}.log
      timeout 300 python bench.py --workload sorted_order --no-cpu-baseline > $O/r6_${TAG}_bench_sorted_order.jsonl 2>> $O/r6_${TAG}.log
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof_default" -o default -- python "$ROOT/bench.py" --no-cpu-baseline --no-robustness --no-through-cpp) > $O/r6_${TAG}_bench_under_rocprof.jsonl 2>> $O/r6_${TAG}.log
      db=$(find $O/prof_default -name "*.db" | head -1)
      [ -n "$db" ] && python scripts/rocprof_summary.py "$db" "round 6 $TAG: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-robustness --no-through-cpp (sort + sorted_order + join + groupby, 5 steps + 2 warm-up each)" | head -60 | cut -c1-190 > $O/r6_${TAG}_default_kernel_stats.txt
      find $O/prof_default -name "*.db" -delete
      pmc() { local wl=$1 ctr=$2; local lc=$(echo $ctr | tr 'A-Z' 'a-z')
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$ROOT/$O/pmc_${wl}_${lc}" -o $wl --output-format csv -- python "$ROOT/bench.py" --workload $wl --rows 1e9 --steps 1 --warmup 0 --no-cpu-baseline) >> $O/r6_${TAG}.log 2>&1; }
      for wl in sort sorted_order join groupby; do pmc $wl FETCH_SIZE; pmc $wl WRITE_SIZE; done
      python scripts/pmc_to_json.py $O $O/r6_pmc_traffic_1e9.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of python bench.py --workload <w> --rows 1e9 --steps 1 --warmup 0 (scripts/gpu_r6.sh $TAG evidence)" | tee $O/r6_${TAG}_pmc_traffic.txt | tail -30
      find $O/pmc_* -name "*.csv" -size +1M -delete
      last_json $O/r6_${TAG}_bench_default.jsonl | cut -c1-1500
      head -24 $O/r6_${TAG}_default_kernel_stats.txt | cut -c1-170 ;;
    *) echo "unknown step $step" ;;
  esac
done
grep -E "Error|error|Traceback" $O/r6_${TAG}.log 2>/dev/null | head -5
exit 0
