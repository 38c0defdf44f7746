#!/bin/bash
# GPU box with ONE GPU: bench.py --gpus 2 at full size with both ranks on cuda:0 (R3N_BENCH_SHARE_GPU=1: gloo for torch, tests/rccl_shim.cpp
# for the library's RCCL calls).  A plumbing run of the driver's launch line -- the two ranks time-share the GPU, the numbers are not scaling.
out=gpurun_out/${1:-two}; mkdir -p $out
lib=$(python -c "import sys; sys.path.insert(0,'tests'); import rccl_shim; print(rccl_shim.build())")
for part in auto objects; do
  for cfg in 3 4; do
    R3N_BENCH_SHARE_GPU=1 R3N_RCCL_LIB=$lib timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 10 --warmup 3 --partition $part --config $cfg > $out/bench_n2_${part}_cfg$cfg.json 2> $out/bench_n2_${part}_cfg$cfg.err
    python - $out/bench_n2_${part}_cfg$cfg.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["n_gpus"], d["config"]["parallelism"][:60], d["config"]["split_model"] and {k:d["config"]["split_model"][k] for k in ("rows_ms","objects_ms","choice")}, d["exchange_ms_per_frame"], d["host_ms_per_frame"])
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
  done
done
