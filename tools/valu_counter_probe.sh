#!/bin/bash
# GPU box: the calibration table of the SQ_INSTS_VALU_* counters (tools/valu_counter_probe.hip).
#   gpurun --timeout 300 -- 'bash tools/valu_counter_probe.sh TAG'
set -u
tag=${1:-probe}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
[ -x tests/_build/valu_counter_probe ] || hipcc --offload-arch=gfx950 -O2 tools/valu_counter_probe.hip -o tests/_build/valu_counter_probe
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 \
  --output-format csv -d "$out/pmc" -o probe -- $root/tests/_build/valu_counter_probe > "$out/probe.log" 2>&1
cd "$root"
python - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
if not f:
    print("no counter file:", open(os.path.join(out, "probe.log")).read()[-800:]); sys.exit(1)
acc = defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"]][r["Counter_Name"]] = acc[r["Kernel_Name"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
N = 4096 * 16
cols = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32"]
lines = ["counter value per executed wave-instruction (one wavefront, %d instructions of one kind per kernel; the prologue's few instructions are the residue)" % N,
         "instruction".ljust(18) + "".join(c.replace("SQ_INSTS_VALU", "VALU").rjust(16) for c in cols)]
for k in sorted(acc):
    name = k.split("(")[0].replace("probe_", "")
    lines.append(name.ljust(18) + "".join(("%.3f" % (acc[k].get(c, 0.0) / N)).rjust(16) for c in cols))
open(os.path.join(out, "valu_counter_probe.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
