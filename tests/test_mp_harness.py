"""CPU: tests/mp_harness.py itself -- the guarantees the GPU suite leans on: a rank that raises, a rank that
dies, a rank that never answers each end the run within its limit with every child reaped and the breadcrumbs in the report."""
import os
import signal
import time

import pytest

import mp_harness


def _ok(rank, world, run_dir, crumb):
    import torch
    dist = mp_harness.init_group(rank, world, run_dir)
    crumb("group up")
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    assert t.item() == world * (world + 1) / 2


def _raises(rank, world, run_dir, crumb):
    crumb("before the failure")
    if rank == 1:
        raise ValueError("rank 1 says no")
    time.sleep(600)  # the peer "waits in a collective"


def _dies(rank, world, run_dir, crumb):
    crumb("about to die" if rank == 0 else "waiting")
    if rank == 0:
        os.kill(os.getpid(), signal.SIGKILL)
    time.sleep(600)


def _hangs(rank, world, run_dir, crumb):
    crumb("stage A")
    crumb("stage B: the last thing anybody hears")
    time.sleep(600)


def _no_children_left():
    import multiprocessing
    return not multiprocessing.active_children()


@pytest.mark.timeout(120)
def test_all_ranks_answer():
    results, problem, rep, s = mp_harness.run_ranks(_ok, 2, limit=90)
    assert problem is None and results == {0: "ok", 1: "ok"} and rep is None, (results, problem, rep)
    assert _no_children_left()


@pytest.mark.timeout(120)
def test_a_rank_that_raises_ends_the_run_in_seconds():
    results, problem, rep, s = mp_harness.run_ranks(_raises, 2, limit=90)
    assert problem is None and set(results) == {1} and "rank 1 says no" in results[1]
    assert s < 60 and "before the failure" in rep
    assert _no_children_left()
    with pytest.raises(pytest.fail.Exception, match="rank 1 says no"):
        mp_harness.check(results, problem, rep, 2)


@pytest.mark.timeout(120)
def test_a_rank_that_dies_ends_the_run_at_once():
    results, problem, rep, s = mp_harness.run_ranks(_dies, 2, limit=90)
    assert problem is not None and "rank 0 died without an answer" in problem and "signal 9" in problem, problem
    assert s < 60 and "about to die" in rep
    assert _no_children_left()


@pytest.mark.timeout(120)
def test_ranks_that_hang_are_dumped_and_killed_within_the_limit():
    results, problem, rep, s = mp_harness.run_ranks(_hangs, 2, limit=25)
    assert not results and problem is not None, (results, problem)
    assert ("no answer within" in problem) or ("died without an answer" in problem), problem  # (the watchdog's exit or the limit: whichever the poll saw first)
    assert s < 45
    assert "stage B: the last thing anybody hears" in rep and "faulthandler" in rep and "_hangs" in rep, rep
    assert _no_children_left()
