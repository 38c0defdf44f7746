"""TEST INFRASTRUCTURE: run a function on `world` spawned processes so that NOTHING can hang the suite.

Why this exists (GPUTEST_r05 rc 124): the multi-process tests waited 900 s on a bare `Queue.get` for ranks that said
nothing, the children were neither inspected nor killed, and the one wait cost the 12 tests behind it and the driver's smoke.

What a run does now:
  * every rank gets a run directory with
      rank<r>.crumbs      one line per stage the WORKER reaches (crumb("...")),
      lib.rank<r>.pid*    one line per stage / collective / host wait the LIBRARY enqueues (R3N_BREADCRUMBS, r3n.hip),
      rank<r>.traceback   faulthandler: every thread's stack if the rank is still alive `limit - 15` s in (then the rank exits),
                          or when it dies on a signal;
  * the rendezvous is a FILE in that directory (no port that is closed and bound again by somebody else);
  * the parent polls the queue AND the children: a rank that dies without an answer ends the wait at once, a rank that reported a
    failure gives its peers five more seconds (they are waiting for it in a collective), the whole run has `limit` seconds;
  * whatever happens every child is terminated, then killed; the failure message carries the tail of every rank's breadcrumbs.
Limit: R3N_MP_LIMIT seconds (default 150; the first process of a fresh box pages torch in for a minute or two)."""
import faulthandler
import glob
import os
import queue
import shutil
import sys
import tempfile
import time
import traceback

LIMIT = float(os.environ.get("R3N_MP_LIMIT", "150"))
HERE = os.path.dirname(os.path.abspath(__file__))


class Crumbs:
    def __init__(self, run_dir, rank):
        self.f = open(os.path.join(run_dir, f"rank{rank}.crumbs"), "a", buffering=1)

    def __call__(self, what):
        self.f.write(f"{time.monotonic():.3f} {what}\n")
        self.f.flush()


def init_group(rank, world, run_dir, device=None, seconds=90):
    """torch.distributed over the run directory's rendezvous file: RCCL when every rank has its own GPU (`device`), else gloo."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # the box's hostname may not resolve
    port = os.environ.get("R3N_MP_TCP_PORT")  # soak only (tools/soak_native.py --rendezvous tcp): the round-5 rendezvous, for comparison
    method = f"tcp://127.0.0.1:{port}" if port else "file://" + os.path.join(run_dir, "rendezvous")
    kw = dict(init_method=method, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=seconds))
    if device is not None:
        dist.init_process_group("nccl", device_id=device, **kw)
    else:
        dist.init_process_group("gloo", **kw)
    return dist


def _entry(target, rank, world, run_dir, q, limit, args, tcp_port=None):
    for p in (HERE, os.path.dirname(HERE)):
        if p not in sys.path:
            sys.path.insert(0, p)
    if tcp_port:
        os.environ["R3N_MP_TCP_PORT"] = str(tcp_port)
    os.environ["R3N_BREADCRUMBS"] = os.path.join(run_dir, f"lib.rank{rank}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    tb = open(os.path.join(run_dir, f"rank{rank}.traceback"), "w")
    faulthandler.enable(file=tb, all_threads=True)
    faulthandler.dump_traceback_later(max(5.0, limit - 15.0), exit=True, file=tb)
    crumb = Crumbs(run_dir, rank)
    crumb(f"started pid {os.getpid()}")
    try:
        target(rank, world, run_dir, crumb, *args)
        crumb("ok")
        q.put((rank, "ok"))
    except BaseException as exc:  # noqa: BLE001
        crumb("FAIL " + repr(exc)[:200])
        q.put((rank, "FAIL: " + repr(exc) + "\n" + traceback.format_exc()))
    finally:
        faulthandler.cancel_dump_traceback_later()
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


def _tail(path, n=12):
    try:
        with open(path, errors="replace") as f:
            lines = f.read().splitlines()
        return lines[-n:]
    except OSError:
        return []


def report(run_dir, world):
    out = []
    for r in range(world):
        out.append(f"--- rank {r}: worker breadcrumbs (tail)")
        out += ["    " + l for l in _tail(os.path.join(run_dir, f"rank{r}.crumbs"), 8)]
        for p in sorted(glob.glob(os.path.join(run_dir, f"lib.rank{r}.pid*"))):
            out.append(f"--- rank {r}: library breadcrumbs {os.path.basename(p)} (tail)")
            out += ["    " + l for l in _tail(p, 10)]
        tb = _tail(os.path.join(run_dir, f"rank{r}.traceback"), 40)
        if tb:
            out.append(f"--- rank {r}: faulthandler")
            out += ["    " + l for l in tb]
    return "\n".join(out)


def _closed_port():
    """What round 5's tests did: bind port 0, read the number, CLOSE the socket, hand the number to the ranks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(target, world, args=(), limit=None, keep=None, rendezvous="file"):
    """-> (results: {rank: "ok" | "FAIL: ..."}, problem: None | str, report: str | None, seconds).  Never raises for a rank's sake,
    never leaves a child behind, never waits longer than `limit` (+ a few seconds of killing)."""
    import torch.multiprocessing as mp
    limit = LIMIT if limit is None else float(limit)
    ctx = mp.get_context("spawn")
    run_dir = tempfile.mkdtemp(prefix="r3n_mp_")
    q = ctx.Queue()
    tcp_port = _closed_port() if rendezvous == "tcp" else None
    procs = [ctx.Process(target=_entry, args=(target, r, world, run_dir, q, limit, tuple(args), tcp_port), daemon=True) for r in range(world)]
    t0 = time.monotonic()
    for p in procs:
        p.start()
    results, problem, first_fail = {}, None, None
    while len(results) < world:
        try:
            rank, msg = q.get(timeout=0.25)
            results[rank] = msg
            if msg != "ok" and first_fail is None:
                first_fail = time.monotonic()
            continue
        except queue.Empty:
            pass
        now = time.monotonic()
        dead = [r for r, p in enumerate(procs) if r not in results and p.exitcode is not None]
        if dead:
            try:  # an answer may still be in the pipe
                while True:
                    rank, msg = q.get(timeout=0.5)
                    results[rank] = msg
            except queue.Empty:
                pass
            dead = [r for r in dead if r not in results]
            if dead:
                problem = "; ".join(f"rank {r} died without an answer (exit code {procs[r].exitcode}"
                                    f"{', signal ' + str(-procs[r].exitcode) if procs[r].exitcode < 0 else ''})" for r in dead)
                break
            continue
        if first_fail is not None and now - first_fail > 5.0:
            break  # the peers of a failed rank wait for it in a collective
        if now - t0 > limit:
            problem = f"no answer within {limit:.0f} s from rank(s) {[r for r in range(world) if r not in results]}"
            break
    for p in procs:
        p.join(timeout=15.0 if (problem is None and first_fail is None) else 0.2)
    for p in procs:
        if p.is_alive():
            p.terminate()
    for p in procs:
        p.join(timeout=3.0)
        if p.is_alive():
            p.kill()
            p.join(timeout=5.0)
    q.close()
    bad = problem is not None or any(m != "ok" for m in results.values()) or len(results) < world
    rep = report(run_dir, world) if bad else None
    keep = os.environ.get("R3N_MP_KEEP") if keep is None else keep
    if bad and keep:
        dst = os.path.join(keep, os.path.basename(run_dir))
        shutil.copytree(run_dir, dst, dirs_exist_ok=True)
    shutil.rmtree(run_dir, ignore_errors=True)
    return results, problem, rep, time.monotonic() - t0


def check(results, problem, rep, world):
    """The assertion every test makes of a run."""
    import pytest
    if problem is not None:
        pytest.fail(problem + "\n" + "\n".join(f"rank {r}: {m}" for r, m in sorted(results.items()) if m != "ok") + "\n" + (rep or ""), pytrace=False)
    failed = {r: m for r, m in results.items() if m != "ok"}
    if failed or len(results) < world:
        pytest.fail("\n".join(f"rank {r}: {m}" for r, m in sorted(failed.items())) +
                    f"\n(ranks without an answer: {[r for r in range(world) if r not in results]})\n" + (rep or ""), pytrace=False)
