set -u
out=gpurun_out/r3e; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_two_process_gpu.py tests/test_scene_viewer.py -m gpu -x -q > $out/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -30 $out/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exchange or sharded or transparent or switches or config4" > $out/pytest_sel.log 2>&1; echo "selected rc=$?"; tail -5 $out/pytest_sel.log
for part in spatial slots; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-exchange --partition $part --steps 40 --warmup 8 > $out/bench_ex_$part.json 2> $out/bench_ex_$part.err; echo "exchange $part rc=$?"
  python - $out/bench_ex_$part.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(d["ms_per_step"], d["exchange_ms_per_frame"], d["exchange_bytes_per_frame"], d["config"]["parallelism"][:60])
except Exception as e: print("FAILED", e)
PY
  tail -3 $out/bench_ex_$part.err
done
timeout 600 python bench.py --config 4 --steps 30 --warmup 6 --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "cfg4 rc=$?"
python - $out/bench_cfg4.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("cfg4", d["ms_per_step"], d["value"], d["culled_objects_per_s"], {k:round(v,4) for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print("FAILED", e)
PY
tail -3 $out/bench_cfg4.err
