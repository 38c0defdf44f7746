# A/B of the variant libraries (twice, to see the run-to-run spread) + the parity tests of the default build
set -u
out=gpurun_out/${1:-ab}; mkdir -p $out
export TMPDIR=/tmp
python tools/variants.py run --steps 150 > $out/variants.txt 2>&1; cat $out/variants.txt
python tools/variants.py run --steps 150 > $out/variants2.txt 2>&1; cat $out/variants2.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
