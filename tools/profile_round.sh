#!/bin/bash
# GPU box: the rocprofv3 passes behind profiles/rNN_summary.md, in the order the task prescribes -- kernel trace + stats
# first, every counter set in its own pass, never mixed with sys / runtime traces.  Output: gpurun_out/$1/.
#   gpurun -- 'bash tools/profile_round.sh r1f'
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
B="python $root/bench.py --no-cpu-baseline"
# 0. the plain bench line of this build (what the summaries are compared with)
(cd "$root" && python bench.py --steps 100 --warmup 10 > "$out/bench.json" 2> "$out/bench.err")
# 1. kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- $B --steps 30 --warmup 5 > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
# 2. HBM traffic, one counter per pass (MI355X_MICROARCH.md, HBM section)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o pmc_fetch -- $B --steps 6 --warmup 2 > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o pmc_write -- $B --steps 6 --warmup 2 > /dev/null 2> "$out/pmc_write.err"
# 3. instruction issue, single stream so that the counters of one kernel are not mixed with its neighbours'
R3N_SINGLE_STREAM=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/sq" -o sq -- $B --steps 6 --warmup 2 > /dev/null 2> "$out/sq.err"
cd "$root"
find "$out" -name "*.csv" -size +20M -delete   # keep the merge-back small: per-dispatch traces of long runs
ls -R "$out" | head -50
# 4. kernel trace without frames in flight: the pass whose per-kernel averages match the bench's HIP-event stage times
cd /tmp
R3N_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_serial" -o kts -- $B --steps 30 --warmup 5 > "$out/bench_under_rocprof_serial.json" 2> "$out/kts.err"
cd "$root"
