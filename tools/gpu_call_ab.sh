# A/B of the variant libraries, twice (run-to-run spread) + a subset of the parity tests on the default build; serial kernel durations under rocprofv3: tools/gpu_call_serial.sh
set -u
out=gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
python tools/variants.py run --steps 150 > $out/variants.txt 2>&1; cat $out/variants.txt
python tools/variants.py run --steps 150 > $out/variants2.txt 2>&1; cat $out/variants2.txt
if [ "${2:-}" = tests ]; then
timeout 900 python -m pytest tests -m gpu -x -q -k "bistro or textured or config or random or golden or msaa or runtime or large or material_key or sharded or two_process" > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E 'passed|failed|error' $out/pytest.log | tail -3; grep -E '^E ' $out/pytest.log | head -5
fi
