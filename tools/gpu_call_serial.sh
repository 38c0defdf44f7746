# Stand-alone kernel durations of the variant libraries: everything on one stream, no frames in flight, under rocprofv3 --stats
set -u
out=$(pwd)/gpurun_out/${1:-ser}; mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for lib in $root/variants/lib_*.so; do
  name=$(basename $lib .so)
  R3N_LIB=$lib R3N_SINGLE_STREAM=1 R3N_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o k -- python $root/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $out/$name.json 2> $out/$name.err
  f=$(find $out/$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; head -12 "$f" | cut -c1-150
  find $out/$name -name "*_kernel_trace.csv" -delete
done
