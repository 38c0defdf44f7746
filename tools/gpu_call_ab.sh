# A/B of the variant libraries, twice (run-to-run spread); serial kernel durations under rocprofv3: tools/gpu_call_serial.sh
set -u
out=gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
python tools/variants.py run --steps 150 > $out/variants.txt 2>&1; cat $out/variants.txt

