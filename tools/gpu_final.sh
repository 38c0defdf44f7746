#!/bin/bash
# Last GPU call(s) of a round on the FINAL kernel sources: the counter passes profiles/traffic.json is made of (bench.py quotes
# counter traffic only from passes taken on the same kernel sources), then the lines that quote them.
#   gpurun --timeout 1500 -- 'bash tools/gpu_final.sh TAG action...'
# actions: tests (all GPU tests) | pmc (default workload, every counter set) | pmc_more (--config 4 and --untextured, traffic only)
#          | bench (headline line with cpu_baseline + parity) | lines (--config 4 with its traffic, --bistro-v2 with parity)
#          | serial (stand-alone kernel durations: default, --config 4, --bistro-v2)
set -u
tag=${1:-r05f}
shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python $root/bench.py"
line() { python - "$1" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
    d=json.loads(l); r=d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "parity", (d.get("parity") or {}).get("ok"), "roofline", r.get("bound"), r.get("frac"), "traffic", r.get("traffic"),
          {k:round(v*1e3,1) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
stats() { f=$(find "$1" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -${2:-16} "$f" | cut -c1-170; find "$1" -name "*_kernel_trace.csv" -delete; }
for w in "$@"; do
  t0=$(date +%s)
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -14 "$out/pytest.log"; grep -E '^E ' "$out/pytest.log" | head -12;;
    pmc) bash tools/gpu_pmc.sh "$tag/pmc" > "$out/pmc_table.txt" 2>&1; tail -3 "$out/pmc_table.txt" | cut -c1-200
      [ -f "$out/traffic.json" ] && cp "$out/traffic.json" "$root/profiles/traffic.json";;
    pmc_more)
      # every counter set for configs[3] and the Bistro-faithful scene too (round 6: a VALU figure for every variant's dominant kernel)
      bash tools/gpu_pmc.sh "$tag/pmc_cfg4" --config 4 > "$out/pmc_table_cfg4.txt" 2>&1
      bash tools/gpu_pmc.sh "$tag/pmc_v2" --bistro-v2 > "$out/pmc_table_v2.txt" 2>&1
      PMC_TRAFFIC_ONLY=1 bash tools/gpu_pmc.sh "$tag/pmc_untextured" --untextured > "$out/pmc_table_untextured.txt" 2>&1
      [ -f "$out/traffic.json" ] && cp "$out/traffic.json" "$root/profiles/traffic.json";;
    bench) $B --steps 100 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"; line "$out/bench.json"; tail -2 "$out/bench.err" | grep -v amdgpu.ids;;
    lines)
      $B --steps 40 --warmup 8 --cpu-sample-frames 1 --config 4 > "$out/bench_cfg4.json" 2> /dev/null; line "$out/bench_cfg4.json"
      cp "$out/bench_cfg4.json" "$out/bench_cfg4_traffic.json"
      $B --steps 60 --warmup 8 --cpu-sample-frames 1 --bistro-v2 > "$out/bench_bistro_v2.json" 2> /dev/null; line "$out/bench_bistro_v2.json";;
    serial)
      cd /tmp
      R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt1" -o kt1 -- $B --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof_single.json" 2> "$out/kt1.err"
      echo "== default, single stream"; stats "$out/kt1" 18
      for v in ${SERIAL_MORE-cfg4 v2}; do  # SERIAL_MORE="" : the default workload only
        flag="--config 4"; [ $v = v2 ] && flag="--bistro-v2"
        R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt1_$v" -o kt1 -- $B --no-cpu-baseline --steps 20 --warmup 5 $flag > "$out/bench_${v}_under_rocprof_single.json" 2> "$out/kt1_$v.err"
        echo "== $v, single stream"; stats "$out/kt1_$v" 14
      done
      cd "$root";;
    *) echo "unknown action $w";;
  esac
  echo "-- $w: $(( $(date +%s) - t0 )) s"
done
find "$out" -name "*.csv" -size +8M -delete
