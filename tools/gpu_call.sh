#!/bin/bash
# One gpurun call of this round: GPU tests, the bench line, the instanced A/B and the two kernel traces.
#   gpurun --timeout 900 -- 'bash tools/gpu_call.sh r2a [tests] [bench] [instanced] [trace]'
set -u
tag=${1:-r2}
shift
what=${*:-tests bench instanced trace}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for w in $what; do
  case $w in
    tests) timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$out/pytest.log";;
    bench) timeout 600 python bench.py --steps 60 --warmup 8 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"; tail -c 3000 "$out/bench.json"; tail -5 "$out/bench.err";;
    instanced) timeout 300 python bench.py --instanced --no-cpu-baseline --steps 60 --warmup 8 > "$out/bench_instanced.json" 2> "$out/bench_instanced.err"; echo "instanced rc=$?"; python -c "
import json,sys
d=json.load(open('$out/bench_instanced.json')); print('instanced', d['ms_per_step'], d['stage_ms_per_frame'])";;
    trace) cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
      R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kts" -o kts -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof_serial.json" 2> "$out/kts.err"
      cd "$root"
      find "$out" -name "*_kernel_trace.csv" -size +20M -delete
      for f in $(find "$out" -name "*kernel_stats.csv"); do echo "== $f"; head -25 "$f" | cut -c1-160; done;;
  esac
done
