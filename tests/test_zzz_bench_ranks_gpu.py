"""GPU: `bench.py --gpus 2` launched exactly as the driver launches it (python -m torch.distributed.run, one process per rank), on a
box with ONE GPU: R3N_BENCH_SHARE_GPU=1 puts both ranks on cuda:0, torch's collectives go over gloo and the library's own over
tests/rccl_shim.cpp (RCCL refuses two ranks on one device).  What is checked is the bench's N > 1 code path itself -- the cost-model
probe and the broadcast of rank 0's choice, r3n_comm_init + the split the library then issues natively, the barrier-bracketed timed
region with the max over ranks, the one JSON line of rank 0 -- not a number: two ranks time-share one GPU here."""
import json
import os
import signal
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, n=2, size=("1280x720", 1280 * 720)):
    sys.path.insert(0, HERE)
    import rccl_shim
    env = dict(os.environ, R3N_BENCH_SHARE_GPU="1", R3N_RCCL_LIB=rccl_shim.build(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "2",
           "--objects", "300", "--tris", "150000", "--resolution", size[0], "--no-cpu-baseline"] + extra
    # its own session: on a time-out the launcher AND every rank are killed (nothing keeps the GPU or the suite)
    limit = float(os.environ.get("R3N_MP_LIMIT", "150")) + 30.0 * (n > 2)
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail(f"bench.py --gpus {n} gave no line within {limit:.0f} s (ranks killed)\n" + out[-2000:] + err[-6000:], pytrace=False)
    assert proc.returncode == 0, out[-2000:] + err[-4000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"rank 0 prints ONE JSON line, got {len(lines)}:\n" + out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("partition", ["auto", "objects", "rows"])
def test_bench_two_ranks_one_line(partition):
    d = _run(["--partition", partition])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2
    assert d["metric"] and d["unit"] and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["scaling"] == "strong" and d["higher_is_better"] is True
    par = d["config"]["parallelism"]
    model = d["config"]["split_model"]
    if partition == "auto":
        assert model["choice"] in ("rows", "objects") and model["rows_ms"] > 0 and model["objects_ms"] > 0, model
        partition = model["choice"]
    else:
        assert model is None
    assert ("object-range split issued by the library" in par) if partition == "objects" else ("sort-first" in par and "issued by the library itself" in par), par
    assert d["exchange_note"] is None, d["exchange_note"]  # (set when r3n_comm_init failed and the torch.distributed fallback ran)
    ex = d["exchange_ms_per_frame"]
    assert ex is not None and set(ex) >= {"shadow", "pass1", "pass2", "rows"}, ex
    assert d["exchange_bytes_per_frame"]["pass2"] == (8 * 1280 * 720 if partition == "objects" else 0)
    assert d["roofline"]["traffic"] is None and "cpu_baseline" not in d  # N = 1 only, as the contract says


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("partition", ["objects", "rows"])
def test_bench_eight_ranks_one_line(partition):
    """the driver's N = 8 launch line on a small target (eight ranks time-share the one GPU: plumbing, not a
    number).  Four shadow views on eight ranks: every view's rows in two bands, one rank each; 360 rows = eight equal bands of 45."""
    d = _run(["--partition", partition], n=8, size=("640x360", 640 * 360))
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["value"] > 0 and d["ms_per_step"] > 0
    par = d["config"]["parallelism"]
    assert ("object-range split issued by the library" in par) if partition == "objects" else ("sort-first" in par), par
    assert d["exchange_note"] is None, d["exchange_note"]
    assert d["exchange_bytes_per_frame"]["pass2"] == (8 * 640 * 360 if partition == "objects" else 0)
