#!/bin/bash
# Round-6 GPU calls.   gpurun --timeout T -- 'bash tools/gpu_r6.sh TAG action...'
#   suite            the driver's round-end sequence in one call: pytest tests -x -q -m gpu, then __graft_entry__.smoke()
#   soak:REPEAT      tools/soak_native.py, every two-rank native* mode REPEAT times        
#   soakmany:REPEAT  the world-4 / world-8 cases REPEAT times
#   soakmode:MODE:REPEAT   one mode
set -u
tag=${1:-r06}
shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for w in "$@"; do
  t0=$(date +%s)
  case $w in
    suite)
      timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -22 "$out/pytest.log" | cut -c1-220
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$out/smoke.log" | cut -c1-220;;
    suiteall)  # the same without -x: every failure of a development run in one call
      timeout 1700 python -m pytest tests -q -m gpu --durations=12 > "$out/pytest_all.log" 2>&1; echo "pytest rc=$?"; tail -30 "$out/pytest_all.log" | cut -c1-260;;
    variants:*)  # variants:NAME:BENCH FLAGS... -- every variants/lib_*.so on the bench (tools/variants.py), log under NAME
      IFS=: read -r _ vname vflags <<< "$w"; VARIANT_FLAGS="$vflags" timeout 1500 python tools/variants.py run --steps 20 > "$out/variants_$vname.txt" 2>&1; cat "$out/variants_$vname.txt" | cut -c1-420;;
    static:*)  # static:SECONDS:FIRST_SEED -- the randomised campaign, three frames per case
      IFS=: read -r _ secs first <<< "$w"; timeout $((secs + 120)) python tools/fuzz_parity.py --seconds "$secs" --first-seed "$first" > "$out/fuzz_static_$first.txt" 2>&1; tail -4 "$out/fuzz_static_$first.txt" | cut -c1-300;;
    mutate:*)  # mutate:SECONDS:FIRST_SEED -- the randomised campaign with world edits (material key flips included)
      IFS=: read -r _ secs first <<< "$w"; timeout $((secs + 120)) python tools/fuzz_parity.py --mutate --seconds "$secs" --first-seed "$first" > "$out/fuzz_mutate_$first.txt" 2>&1; tail -4 "$out/fuzz_mutate_$first.txt" | cut -c1-300;;
    asyncsoak:*)  # asyncsoak:SERIAL:REPEAT -- the asynchronous shim with comm_serial = SERIAL, two-rank and many-rank cases
      IFS=: read -r _ ser rep <<< "$w"
      timeout 1500 python tools/soak_native.py --shim async --serial "$ser" --repeat "$rep" --limit 90 --out "$out/async_serial$ser" > "$out/async_serial${ser}_stdout.log" 2>&1; echo "async serial=$ser rc=$?"; tail -2 "$out/async_serial${ser}_stdout.log" | cut -c1-500
      timeout 1500 python tools/soak_native.py --shim async --serial "$ser" --many --repeat "$rep" --limit 90 --out "$out/async_many_serial$ser" > "$out/async_many_serial${ser}_stdout.log" 2>&1; echo "async many serial=$ser rc=$?"; tail -2 "$out/async_many_serial${ser}_stdout.log" | cut -c1-500;;
    bench:*)  # bench:NAME:FLAGS -- one bench.py line, stage table printed
      IFS=: read -r _ bname bflags <<< "$w"; timeout 1200 python bench.py $bflags > "$out/bench_$bname.json" 2> "$out/bench_$bname.err"; python - "$out/bench_$bname.json" <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
    d=json.loads(l); r=d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "parity", (d.get("parity") or {}).get("ok"), "roofline", r.get("kernel"), r.get("bound"), r.get("frac"), "traffic", r.get("traffic"),
          {k:round(v*1e3,1) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
      tail -3 "$out/bench_$bname.err" | grep -v amdgpu.ids | cut -c1-300;;
    pytest:*)  # pytest:NAME:ARGS -- a selection of the GPU tests
      IFS=: read -r _ pname pargs <<< "$w"; timeout 1500 python -m pytest $pargs -q -m gpu > "$out/pytest_$pname.log" 2>&1; echo "pytest $pname rc=$?"; tail -6 "$out/pytest_$pname.log" | cut -c1-300;;
    soaktcp:*) timeout 3000 python tools/soak_native.py --rendezvous tcp --repeat "${w#soaktcp:}" --limit 100 --out "$out/soak_tcp" > "$out/soak_tcp_stdout.log" 2>&1; echo "soak tcp rc=$?"; tail -3 "$out/soak_tcp_stdout.log" | cut -c1-500;;
    soak:*) timeout 3000 python tools/soak_native.py --repeat "${w#soak:}" --out "$out/soak" > "$out/soak_stdout.log" 2>&1; echo "soak rc=$?"; tail -4 "$out/soak_stdout.log" | cut -c1-400;;
    soakmany:*) timeout 3000 python tools/soak_native.py --many --repeat "${w#soakmany:}" --out "$out/soak_many" > "$out/soak_many_stdout.log" 2>&1; echo "soak many rc=$?"; tail -4 "$out/soak_many_stdout.log" | cut -c1-400;;
    soakmode:*) IFS=: read -r _ mode rep <<< "$w"; timeout 3000 python tools/soak_native.py --modes "$mode" --repeat "$rep" --out "$out/soak_$mode" > "$out/soak_${mode}_stdout.log" 2>&1; echo "soak $mode rc=$?"; tail -4 "$out/soak_${mode}_stdout.log" | cut -c1-400;;
    *) echo "unknown action $w";;
  esac
  echo "-- $w: $(( $(date +%s) - t0 )) s"
done
