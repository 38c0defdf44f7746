#!/bin/bash
# Round-4 GPU call: several actions in one box acquisition.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r4.sh TAG action...'
# actions: quick (subset of the parity tests) | tests (all GPU tests) | bench | classes (general kernel vs material classes) |
#          variants (tools/variants.py run x2) | serial (stand-alone kernel durations of every variant library) |
#          pmc_resolve | pmc_raster (stall / memory-side counter sets of the rasteriser kernels, one set per rocprofv3 run)
set -u
tag=${1:-r4}
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
pmc() {  # name, env, counters...
  local name=$1 envs=$2; shift 2
  ( cd /tmp; env $envs timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$name" -o "$name" -- python $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> "$out/pmc_$name.err" ) || echo "pmc $name failed: $(tail -2 $out/pmc_$name.err)"
  find "$out/pmc_$name" -name "*_kernel_trace.csv" -delete
}
for w in "$@"; do
  case $w in
    quick) timeout 900 python -m pytest tests -m gpu -x -q -k "textured or bistro or runtime or golden or float_textures or encoded or vertex_colour or material_key or frames_in_flight or shade_mode" > "$out/pytest_quick.log" 2>&1; echo "pytest quick rc=$?"; tail -4 "$out/pytest_quick.log"; grep -E '^E ' "$out/pytest_quick.log" | head -8;;
    shim) timeout 900 python -m pytest tests/test_zzz_multi_process_gpu.py -q -x > "$out/pytest_shim.log" 2>&1; echo "pytest two-process rc=$?"; tail -6 "$out/pytest_shim.log"; grep -E '^E |FAIL' "$out/pytest_shim.log" | head -12;;
    trace) R3N_LIB=$root/variants/lib_trace.so R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 400 python tools/wave_trace.py > "$out/wave_trace_single.txt" 2>&1; grep -A12 "per-triangle pass, quadrant [01]" "$out/wave_trace_single.txt" | cut -c1-400
           R3N_LIB=$root/variants/lib_trace.so timeout 400 python tools/wave_trace.py > "$out/wave_trace.txt" 2>&1; grep -A3 "per-triangle pass, quadrant 0" "$out/wave_trace.txt" | cut -c1-400;;
    probe) timeout 300 python tools/exact_math_probe.py > "$out/exact_math.txt" 2>&1; cat "$out/exact_math.txt" | head -80;;
    quick_lean2) R3N_LIB=$root/variants/lib_lean2.so timeout 900 python -m pytest tests -m gpu -x -q -k "bistro or random or golden or config4_full or near_plane or large_scene or msaa_random or transparent" > "$out/pytest_lean2.log" 2>&1; echo "pytest lean2 rc=$?"; tail -4 "$out/pytest_lean2.log"; grep -E '^E ' "$out/pytest_lean2.log" | head -8;;
    bisect)
      for lib in $root/variants/lib_*.so; do
        name=$(basename $lib .so)
        R3N_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_exact_math.py -m gpu -q -k "static_gltf or config3_4k or exact_math or config4_full or transparent or msaa_random or vertex_colour" > "$out/pytest_$name.log" 2>&1
        echo "== $name: $(tail -1 $out/pytest_$name.log)"; grep -E '^E  +AssertionError' "$out/pytest_$name.log" | cut -c1-200
      done;;
    quick_texf) R3N_LIB=$root/variants/lib_texf.so timeout 900 python -m pytest tests -m gpu -x -q -k "textured or bistro or golden_static or float_textures or encoded or scene_viewer" > "$out/pytest_texf.log" 2>&1; echo "pytest texf rc=$?"; tail -4 "$out/pytest_texf.log"; grep -E '^E ' "$out/pytest_texf.log" | head -8;;
    tests) timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$out/pytest.log"; grep -E '^E ' "$out/pytest.log" | head -8;;
    bench) python bench.py --steps 100 --warmup 10 > "$out/bench.json" 2> "$out/bench.err"; line "$out/bench.json"; tail -3 "$out/bench.err";;
    classes)
      for rep in 1 2; do
        R3N_RESOLVE_CLASSES=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$out/bench_general_$rep.json" 2>/dev/null; line "$out/bench_general_$rep.json"
        python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "$out/bench_classes_$rep.json" 2>/dev/null; line "$out/bench_classes_$rep.json"
      done
      python bench.py --steps 100 --warmup 10 --no-cpu-baseline --shade-mode fast > "$out/bench_fast.json" 2>/dev/null; line "$out/bench_fast.json"
      python bench.py --steps 100 --warmup 10 --no-cpu-baseline --untextured > "$out/bench_untextured.json" 2>/dev/null; line "$out/bench_untextured.json";;
    variants) python tools/variants.py run --steps 120 > "$out/variants.txt" 2>&1; cat "$out/variants.txt"; python tools/variants.py run --steps 120 > "$out/variants2.txt" 2>&1; cat "$out/variants2.txt";;
    serial)
      cd /tmp
      for lib in $root/variants/lib_*.so; do
        name=$(basename $lib .so)
        R3N_LIB=$lib R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_$name -o k -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $out/ser_$name.json 2> $out/ser_$name.err
        f=$(find $out/ser_$name -name "*kernel_stats.csv" | head -1)
        echo "== $name"; head -14 "$f" | cut -c1-150
        find $out/ser_$name -name "*_kernel_trace.csv" -delete
      done
      cd $root;;
    pmc_resolve)
      pmc rw "R3N_SINGLE_STREAM=1" WRITE_SIZE
      pmc rf "R3N_SINGLE_STREAM=1" FETCH_SIZE
      pmc rsq "R3N_SINGLE_STREAM=1" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
      pmc rmix "R3N_SINGLE_STREAM=1" SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM_RD
      pmc rw0 "R3N_SINGLE_STREAM=1 R3N_RESOLVE_CLASSES=0" WRITE_SIZE
      pmc rsq0 "R3N_SINGLE_STREAM=1 R3N_RESOLVE_CLASSES=0" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
      python tools/pmc_table.py $(find $out/pmc_r* -name "*counter_collection.csv") 2>/dev/null | grep -E "kernel|k_resolve|---" | cut -c1-300;;
    pmc_raster)
      S="R3N_SINGLE_STREAM=1 R3N_PIPELINE=0"
      pmc a1 "$S" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
      pmc a2 "$S" SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
      pmc a3 "$S" SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU
      pmc a4 "$S" TCP_PENDING_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum
      pmc a5 "$S" TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
      pmc a6 "$S" TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_ATOMIC_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum TA_TA_BUSY_sum
      pmc a7 "$S" TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
      pmc a8 "$S" TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
      pmc a9 "$S" TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum
      pmc a10 "$S" SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES
      pmc a11 "$S" TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_ATOMIC_32B_sum TCC_BUSY_sum TCC_CYCLE_sum
      python tools/pmc_table.py $(find $out/pmc_a* -name "*counter_collection.csv") > "$out/pmc_raster_table.md" 2>/dev/null
      grep -E "kernel|k_raster|k_triangle|---" "$out/pmc_raster_table.md" | cut -c1-600;;
  esac
done
find "$out" -name "*.csv" -size +16M -delete
