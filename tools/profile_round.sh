#!/bin/bash
# GPU box: everything behind profiles/rNN_*: bench lines of the build (default / fast / instanced / untextured / MSAA / config 4 /
# a real asset through the scene-viewer harness / the multi-GPU exchange forced at N = 1 for every split), host cost per frame,
# rocprofv3 kernel traces (frames in flight and serial), the PMC passes (one counter set per run, kernel-trace only), the other
# BASELINE.json configs.   gpurun --timeout 1800 -- 'bash tools/profile_round.sh r03'
set -u
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python $root/bench.py"
$B --steps 100 --warmup 10 --no-cpu-baseline --shade-mode fast > "$out/bench_fast.json" 2>/dev/null
$B --steps 100 --warmup 10 --no-cpu-baseline --instanced > "$out/bench_instanced.json" 2>/dev/null
$B --steps 100 --warmup 10 --no-cpu-baseline --untextured > "$out/bench_untextured.json" 2>/dev/null
$B --steps 60 --warmup 10 --no-cpu-baseline --samples 4 > "$out/bench_msaa4.json" 2>/dev/null
$B --steps 40 --warmup 8 --cpu-sample-frames 1 --config 4 > "$out/bench_cfg4.json" 2>/dev/null   # with the oracle's parity block (one sample frame)
$B --steps 60 --warmup 8 --cpu-sample-frames 1 --bistro-v2 > "$out/bench_bistro_v2.json" 2>/dev/null   # the Bistro-faithful stand-in, with the oracle's parity block
$B --steps 60 --warmup 8 --cpu-sample-frames 3 --scene tests/golden/static_gltf-data.glb --directional-light=-1,-4,2 --directional-light-intensity 4 --shadow-distance 20 --camera=3,3,5,-0.55,-0.5 > "$out/bench_scene.json" 2>/dev/null
for part in rows objects rows_python spatial slots; do
  flags="--partition $part"; [ $part = rows_python ] && flags="--partition rows --python-exchange"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-exchange $flags --steps 60 --warmup 8 2>/dev/null | grep '^{' > "$out/bench_exchange_$part.json"
done
python tools/host_rate.py > "$out/host_rate.txt" 2>&1
for sc in default cfg4 v2; do python tools/frame_timeline.py --scene $sc 2>&1 | grep -v amdgpu > "$out/frame_timeline_$sc.txt"; done   # the schedule of the pipelined multi-stream frame (the library's own timing taps)
R3N_FRAME_NODES=1 python tools/host_rate.py > "$out/host_rate_nodes.txt" 2>&1
python tools/run_config.py cfg2 cfg4 cfg5 cfg5anim cfg5asset > "$out/configs.jsonl" 2> "$out/configs.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- $B --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
R3N_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kts" -o kts -- $B --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof_serial.json" 2> "$out/kts.err"
# everything on ONE stream, no frames in flight: a kernel's average duration here is its stand-alone time -- what the line's
# rooflines.*.ms_per_launch (HIP events, single stream) must agree with
R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt1" -o kt1 -- $B --no-cpu-baseline --steps 30 --warmup 5 > "$out/bench_under_rocprof_single.json" 2> "$out/kt1.err"
R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt1_cfg4" -o kt1 -- $B --no-cpu-baseline --steps 20 --warmup 5 --config 4 > "$out/bench_cfg4_under_rocprof_single.json" 2> "$out/kt1_cfg4.err"
R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt1_v2" -o kt1 -- $B --no-cpu-baseline --steps 20 --warmup 5 --bistro-v2 > "$out/bench_v2_under_rocprof_single.json" 2> "$out/kt1_v2.err"
python $root/tools/exact_math_probe.py > "$out/exact_math.txt" 2>&1
cd "$root"
# PMC passes: every counter set for the default workload; FETCH_SIZE / WRITE_SIZE alone for the other variants whose lines quote
# counter traffic (--config 4: VERDICT r4 item 8) and for the factor-only variant (resolve's texel share = textured - untextured).
# tools/make_traffic.py merges the variants into $out/traffic.json (one entry per workload variant).
bash tools/gpu_pmc.sh "$tag/pmc" > "$out/pmc_table.txt" 2>&1
PMC_TRAFFIC_ONLY=1 bash tools/gpu_pmc.sh "$tag/pmc_cfg4" --config 4 > "$out/pmc_table_cfg4.txt" 2>&1
PMC_TRAFFIC_ONLY=1 bash tools/gpu_pmc.sh "$tag/pmc_untextured" --untextured > "$out/pmc_table_untextured.txt" 2>&1
# the headline lines LAST, quoting the counter traffic of THIS build (bench.py refuses traffic.json of other kernel sources)
[ -f "$out/traffic.json" ] && cp "$out/traffic.json" "$root/profiles/traffic.json"
$B --steps 40 --warmup 8 --no-cpu-baseline --config 4 > "$out/bench_cfg4_traffic.json" 2>/dev/null   # (the same line as bench_cfg4 with its counter traffic)
$B --steps 100 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"
# (round 4's eleven stall / memory-side counter passes of the rasterisers: tools/gpu_r4.sh pmc_raster, not repeated every round)
[ "${WITH_STALL:-0}" = 1 ] && bash tools/gpu_r4.sh "$tag/stall" pmc_raster > "$out/pmc_raster.txt" 2>&1
find "$out" -name "*_kernel_trace.csv" -size +8M -delete
find "$out" -name "*counter_collection.csv" -size +8M -delete
ls "$out"
for f in bench bench_fast bench_instanced bench_untextured bench_msaa4 bench_cfg4 bench_cfg4_traffic bench_bistro_v2 bench_scene bench_exchange_rows bench_exchange_objects bench_exchange_rows_python bench_exchange_spatial bench_exchange_slots; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    line=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
    d=json.loads(line); print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], "host", d.get("host_ms_per_frame"), "parity", (d.get("parity") or {}).get("ok"), {k:round(v,3) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
head -2 "$out/host_rate.txt" "$out/host_rate_nodes.txt" | grep -v amdgpu
cat "$out/configs.jsonl" | cut -c1-400
