# the multi-GPU path on one box: two-process tests, the sharding tests, the exchange forced with one rank under RCCL
set -u
out=gpurun_out/${1:-ex}; mkdir -p $out
export TMPDIR=/tmp
if [ "${2:-tests}" = tests ]; then
timeout 900 python -m pytest tests -m gpu -x -q -k "two_process or exchange or sharded or native_comm" > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E 'passed|failed|error' $out/pytest.log | tail -3; grep -E '^E ' $out/pytest.log | head -20
fi
for part in "rows" "rows --python-exchange" "slots"; do
  tag=$(echo $part | tr -d ' -')
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-exchange --partition $part --steps 60 --warmup 8 > $out/bench_ex_$tag.json 2> $out/bench_ex_$tag.err; echo "exchange $part rc=$?"
  python - $out/bench_ex_$tag.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(d["ms_per_step"], "host", d.get("host_ms_per_frame"), d.get("exchange_ms_per_frame"), d.get("exchange_bytes_per_frame"))
except Exception as e: print("FAILED", e)
PY
  tail -3 $out/bench_ex_$tag.err | cut -c1-300
done
timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 8 > $out/bench_plain.json 2>/dev/null; python -c "
import json;d=json.load(open('$out/bench_plain.json'));print('plain',d['ms_per_step'],'host',d['host_ms_per_frame'])"
