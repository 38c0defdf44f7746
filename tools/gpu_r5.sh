#!/bin/bash
# Round-5 GPU call: several actions in one box acquisition.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5.sh TAG action...'
# actions: quick (parity subset) | tests (all GPU tests) | bench (default + config 4 lines) | serial (stand-alone kernel durations,
#          default + config 4) | variants (tools/variants.py run x2) | vserial (stand-alone kernel durations of every variant library)
set -u
tag=${1:-r5}
shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
    d=json.loads(l); print(sys.argv[1].split('/')[-1], d["ms_per_step"], "parity", (d.get("parity") or {}).get("ok"), {k:round(v*1e3,1) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
stats() {  # dir: the top of the kernel stats csv
  f=$(find "$1" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -${2:-16} "$f" | cut -c1-170
  find "$1" -name "*_kernel_trace.csv" -delete
}
for w in "$@"; do
  case $w in
    quick) timeout 1200 python -m pytest tests -m gpu -x -q -k "${QUICK_K:-golden or random_scene or config4 or config3 or bistro or runtime or material_key or frames_in_flight or transparent or sharded or large_scene or empty_and}" > "$out/pytest_quick.log" 2>&1; echo "pytest quick rc=$?"; tail -4 "$out/pytest_quick.log"; grep -E '^E ' "$out/pytest_quick.log" | head -12;;
    tests) timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$out/pytest.log"; grep -E '^E ' "$out/pytest.log" | head -12;;
    bench)
      python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; line "$out/bench.json"; tail -2 "$out/bench.err" | grep -v amdgpu.ids
      python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$out/bench_b.json" 2> /dev/null; line "$out/bench_b.json"
      python bench.py --steps 40 --warmup 8 --no-cpu-baseline --config 4 > "$out/bench_cfg4.json" 2> "$out/bench_cfg4.err"; line "$out/bench_cfg4.json"; tail -2 "$out/bench_cfg4.err" | grep -v amdgpu.ids;;
    benchp) python bench.py --steps 40 --warmup 8 --cpu-sample-frames 1 --config 4 > "$out/bench_cfg4_parity.json" 2> "$out/bench_cfg4_parity.err"; line "$out/bench_cfg4_parity.json";;
    serial)
      cd /tmp
      R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_default -o k -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $out/ser_default.json 2> $out/ser_default.err
      echo "== default, single stream"; stats $out/ser_default 18
      R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_cfg4 -o k -- python $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 --config 4 > $out/ser_cfg4.json 2> $out/ser_cfg4.err
      echo "== config 4, single stream"; stats $out/ser_cfg4 18
      cd $root;;
    variants) python tools/variants.py run --steps 120 > "$out/variants.txt" 2>&1; cat "$out/variants.txt"; python tools/variants.py run --steps 120 > "$out/variants2.txt" 2>&1; cat "$out/variants2.txt";;
    vserial)
      cd /tmp
      for lib in $root/variants/lib_*.so; do
        name=$(basename $lib .so)
        R3N_LIB=$lib R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_$name -o k -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 ${VSERIAL_FLAGS:-} > $out/ser_$name.json 2> $out/ser_$name.err
        echo "== $name"; stats $out/ser_$name 14
      done
      cd $root;;
    envab)  # A/B of a run-time switch inside one call: ENVAB="R3N_X=0 R3N_X=1" (each setting: two frame runs + the stand-alone kernel durations)
      for setting in ${ENVAB:-R3N_SHADOW_TILE_LDS=0 R3N_SHADOW_TILE_LDS=1}; do
        for rep in 1 2; do
          env $setting python bench.py --steps 100 --warmup 10 --no-cpu-baseline ${ENVAB_FLAGS:-} > "$out/bench_${setting}_$rep.json" 2> "$out/bench_${setting}_$rep.err"; line "$out/bench_${setting}_$rep.json"
        done
      done
      cd /tmp
      for setting in ${ENVAB:-R3N_SHADOW_TILE_LDS=0 R3N_SHADOW_TILE_LDS=1}; do
        env $setting R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_$setting -o k -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 ${ENVAB_FLAGS:-} > $out/ser_$setting.json 2> $out/ser_$setting.err
        echo "== $setting, single stream"; stats $out/ser_$setting 14 | grep -v "k_copy\|copyBuffer"
      done
      cd $root;;
    v2)  # the Bistro-faithful stand-in: its parity test, its bench line with the oracle's parity block, its stand-alone kernel durations
      timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bistro_v2" > "$out/pytest_v2.log" 2>&1; echo "pytest v2 rc=$?"; tail -3 "$out/pytest_v2.log"; grep -E '^E ' "$out/pytest_v2.log" | head -8
      python bench.py --bistro-v2 --steps 60 --warmup 8 --cpu-sample-frames 1 > "$out/bench_bistro_v2.json" 2> "$out/bench_bistro_v2.err"; line "$out/bench_bistro_v2.json"; tail -2 "$out/bench_bistro_v2.err" | grep -v amdgpu.ids
      cd /tmp
      R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_v2 -o k -- python $root/bench.py --bistro-v2 --no-cpu-baseline --steps 30 --warmup 5 > $out/ser_v2.json 2> $out/ser_v2.err
      echo "== bistro v2, single stream"; stats $out/ser_v2 20 | grep -v "k_copy\|copyBuffer"
      cd $root;;
    *) echo "unknown action $w";;
  esac
done
find "$out" -name "*.csv" -size +16M -delete
