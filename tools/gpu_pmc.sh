#!/bin/bash
# PMC passes of the bench (each counter set in its own rocprofv3 run, kernel-trace only -- never with sys / runtime traces).
#   gpurun --timeout 900 -- 'bash tools/gpu_pmc.sh r2pmc [extra bench args]'
set -u
tag=${1:-pmc}
shift
extra="$*"
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
B="python $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 $extra"
rocprofv3 -L > "$out/counters_avail.txt" 2>&1
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST[A-Z_0-9]*" "$out/counters_avail.txt" | sort -u | tr '\n' ' ' | head -c 3000; echo
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- $B > /dev/null 2> "$out/$name.err" || echo "$name failed: $(tail -2 $out/$name.err)"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
if [ "${PMC_TRAFFIC_ONLY:-0}" != 1 ]; then   # (the other workload variants: the two traffic counters only)
R3N_SINGLE_STREAM=1 run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
R3N_SINGLE_STREAM=1 run ic SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
R3N_SINGLE_STREAM=1 run mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM_RD
R3N_SINGLE_STREAM=1 run mix2 SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VSKIPPED SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_IOPS
fi
cd "$root"
python tools/make_traffic.py "$out" $extra
find "$out" -name "*.csv" -size +20M -delete
