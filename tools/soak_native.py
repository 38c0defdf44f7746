#!/usr/bin/env python3
"""Soak of the library's own multi-rank exchange: the workers of tests/test_zzz_multi_process_gpu.py launched
over and over on one lease through tests/mp_harness.py (file rendezvous, worker + library breadcrumbs, faulthandler, children
always reaped), one line per launch, the run directory of every launch that failed kept under <out>/failed/.

    python tools/soak_native.py --repeat 60 --out gpurun_out/soak                 # every two-rank native* mode, 60 times each
    python tools/soak_native.py --repeat 10 --many --out gpurun_out/soak_many     # the world-4 / world-8 cases
    python tools/soak_native.py --repeat 40 --modes native-msaa --limit 90

A "launch" = one spawn of `world` fresh processes: torch + the library loaded, two contexts each, the communicators made, four
frames (camera moving) rendered sharded and unsharded and compared bit for bit, everything torn down."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=20)
    ap.add_argument("--modes", nargs="*", default=None)
    ap.add_argument("--many", action="store_true", help="the world-4 / world-8 parametrisations instead of the two-rank ones")
    ap.add_argument("--limit", type=float, default=120.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak"))
    ap.add_argument("--max-failures", type=int, default=6)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop launching after this many seconds (0: no limit)")
    ap.add_argument("--rendezvous", default="file", choices=("file", "tcp"), help="tcp: round 5's rendezvous (a port number taken from a socket that is closed again), for comparison")
    ap.add_argument("--shim", default="sync", choices=("sync", "async"), help="tests/rccl_shim.cpp's mode (async: collectives run when their stream reaches them)")
    ap.add_argument("--serial", type=int, default=1, choices=(0, 1), help="the library's comm_serial tunable (0: the communicators' collectives unordered against each other)")
    a = ap.parse_args()
    import mp_harness
    import test_zzz_multi_process_gpu as T
    os.makedirs(os.path.join(a.out, "failed"), exist_ok=True)
    if a.many:
        cases = [(w, m, n) for (w, m, n) in T.MANY if a.modes is None or m in a.modes]
    else:
        cases = [(2, m, 200) for m in (a.modes if a.modes is not None else [m for m in T.MODES if m.startswith("native")])]
    log = open(os.path.join(a.out, "soak.log"), "a", buffering=1)
    t_start, n, failures, secs = time.monotonic(), 0, [], []
    for it in range(a.repeat):
        for world, mode, n_objects in cases:
            if a.seconds and time.monotonic() - t_start > a.seconds:
                break
            results, problem, rep, s = mp_harness.run_ranks(T.worker, world, T.worker_args(mode, world, n_objects, a.shim, a.serial), limit=a.limit,
                                                            keep=os.path.join(a.out, "failed"), rendezvous=a.rendezvous)
            ok = problem is None and len(results) == world and all(m == "ok" for m in results.values())
            n += 1
            secs.append(s)
            line = f"{n:5d} it {it:3d} world {world} {mode:24s} {s:6.1f} s {'ok' if ok else 'FAILED'}"
            print(line, flush=True)
            log.write(line + "\n")
            if not ok:
                detail = (problem or "") + "\n" + "\n".join(f"rank {r}: {m}" for r, m in sorted(results.items()) if m != "ok") + "\n" + (rep or "")
                log.write(detail + "\n")
                print(detail[-6000:], flush=True)
                failures.append({"launch": n, "world": world, "mode": mode, "problem": problem, "seconds": s})
                if len(failures) >= a.max_failures:
                    break
        else:
            continue
        break
    summary = {"launches": n, "failed": len(failures), "failures": failures, "seconds_total": round(time.monotonic() - t_start, 1),
               "seconds_per_launch_mean": round(sum(secs) / max(1, len(secs)), 2), "seconds_per_launch_max": round(max(secs or [0]), 2),
               "cases": [f"{w}:{m}" for w, m, _ in cases], "limit_s": a.limit, "shim": a.shim, "comm_serial": a.serial, "rendezvous": a.rendezvous}
    print(json.dumps(summary))
    with open(os.path.join(a.out, "soak_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
