#!/bin/bash
# GPU box: vector / scalar instruction counts of the resolve (per launch) for every variants/lib_*.so -- one rocprofv3 --pmc pass each
# (tools/variants.py build ... first).  usage: bash tools/pmc_variants.sh TAG
set -u
root=$(pwd); out=$root/gpurun_out/${1:-pmcv}; mkdir -p $out; export TMPDIR=/tmp
for lib in $root/variants/lib_*.so; do
  name=$(basename $lib .so)
  ( cd /tmp; R3N_LIB=$lib R3N_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/$name -o p -- python $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> $out/$name.err )
  find $out/$name -name "*_kernel_trace.csv" -delete
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  python - "$f" "$name" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "k_resolve_opaque" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,c in acc.items():
    L=max(n[(k,x)] for x in c)
    px=3840*2160
    print(sys.argv[2], k[:40], "launches",L, {x: round(v/L/ (px/64) ,1) for x,v in c.items() if x.startswith("SQ_INSTS")}, "per pixel-wave;", {x: round(v/L) for x,v in c.items() if not x.startswith("SQ_INSTS")})
PY
done
