set -u
out=gpurun_out/r3k; mkdir -p $out
export TMPDIR=/tmp
python tools/variants.py run --steps 200 > $out/variants.txt 2>&1; cat $out/variants.txt
python tools/variants.py run --steps 200 > $out/variants2.txt 2>&1; cat $out/variants2.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hiz or config3 or random_scene or msaa" 2>&1 | tail -3
cd /tmp
R3N_LIB=$GRAFT_REPO_ROOT/variants/lib_base.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/kt_base -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 60 --warmup 5 > /dev/null 2>&1
R3N_LIB=$GRAFT_REPO_ROOT/variants/lib_neither.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/kt_neither -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 60 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $out -name "*_kernel_trace.csv" -delete
for v in base neither; do echo "== $v"; f=$(find $out/kt_$v -name "*kernel_stats.csv" | head -1); grep -i "hiz\|triangle_cull\|raster_big<false\|raster_small<false\|chained" $f | cut -c1-150; done
