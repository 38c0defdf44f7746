#!/bin/bash
# GPU box: everything behind profiles/rNN_*: bench lines of the build (default / fast / instanced / MSAA / tile-owned shadows),
# rocprofv3 kernel traces (frames in flight and serial), the PMC passes (one counter set per run, kernel-trace only), the other
# BASELINE.json configs.   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
set -u
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python $root/bench.py"
$B --steps 100 --warmup 10 --no-cpu-baseline --shade-mode fast > "$out/bench_fast.json" 2>/dev/null
$B --steps 100 --warmup 10 --no-cpu-baseline --instanced > "$out/bench_instanced.json" 2>/dev/null
$B --steps 100 --warmup 10 --no-cpu-baseline --untextured > "$out/bench_untextured.json" 2>/dev/null
$B --steps 60 --warmup 10 --no-cpu-baseline --samples 4 > "$out/bench_msaa4.json" 2>/dev/null
R3N_SHADOW_TILES=1 $B --steps 100 --warmup 10 --no-cpu-baseline > "$out/bench_shadow_tiles.json" 2>/dev/null
python tools/run_config.py cfg2 cfg4 cfg5 cfg5mfma cfg5anim cfg5asset > "$out/configs.jsonl" 2> "$out/configs.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- $B --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
R3N_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kts" -o kts -- $B --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof_serial.json" 2> "$out/kts.err"
cd "$root"
bash tools/gpu_pmc.sh "$tag/pmc" > "$out/pmc_table.txt" 2>&1
bash tools/gpu_pmc.sh "$tag/pmc_fast" --shade-mode fast > "$out/pmc_fast_table.txt" 2>&1
# the headline line LAST, quoting the counter traffic of THIS build (bench.py refuses traffic.json of other kernel sources)
[ -f "$out/pmc/traffic.json" ] && cp "$out/pmc/traffic.json" "$root/profiles/traffic.json"
$B --steps 100 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"
find "$out" -name "*_kernel_trace.csv" -size +8M -delete
find "$out" -name "*counter_collection.csv" -size +8M -delete
ls "$out"
for f in bench bench_fast bench_instanced bench_untextured bench_msaa4 bench_shadow_tiles; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], d.get("parity",{}).get("ok"), {k:round(v,3) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
cat "$out/configs.jsonl" | cut -c1-400
