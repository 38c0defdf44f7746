#!/bin/bash
# One gpurun call: GPU tests, the bench line, host enqueue rate, kernel traces.
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh r3a [tests] [bench] [host] [trace] [variants]'
set -u
tag=${1:-r3}
shift
what=${*:-tests bench host}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for w in $what; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -25 "$out/pytest.log";;
    bench) R3N_VERBOSE=1 timeout 600 python bench.py --steps 60 --warmup 8 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"; python - "$out/bench.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("bench", d["ms_per_step"], d["value"], "parity", d.get("parity",{}).get("ok"), "hbm", d.get("hbm_copy_rate_measured_GBps"), {k:round(v,4) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print("bench FAILED", e)
PY
      tail -5 "$out/bench.err";;
    host) timeout 300 python tools/host_rate.py > "$out/host_rate.txt" 2>&1; head -3 "$out/host_rate.txt"; sed -n 4,30p "$out/host_rate.txt" | cut -c1-150
      R3N_FRAME_NODES=1 timeout 300 python tools/host_rate.py > "$out/host_rate_nodes.txt" 2>&1; head -1 "$out/host_rate_nodes.txt";;
    variants) timeout 900 python tools/variants.py run --steps 60 > "$out/variants.txt" 2>&1; cat "$out/variants.txt";;
    trace) cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
      R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kts" -o kts -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof_serial.json" 2> "$out/kts.err"
      cd "$root"
      find "$out" -name "*_kernel_trace.csv" -size +20M -delete
      for f in $(find "$out" -name "*kernel_stats.csv"); do echo "== $f"; head -25 "$f" | cut -c1-160; done;;
  esac
done
