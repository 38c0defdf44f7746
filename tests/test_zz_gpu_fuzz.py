"""GPU: a fixed slice of the randomised parity campaign (tools/fuzz_parity.py) inside the suite.

Every case draws its scene, target and (in the mutating mode) its world edits from its seed and compares every frame of the HIP path
with the oracle like the hand-written tests do (compare_frames: sets, keys, atlas, HDR bit-exact).  The campaign proper runs for
minutes with fresh seeds (profiles/r05_fuzz_parity*.txt: 6 377 cases after the two defects it found were fixed); the seeds below
are the ranges in which those defects first showed -- an object added to an exactly full object buffer (static mode, third frame),
a target resized between frames (mutating mode) -- so that a regression fails here by seed.
(The file sorts last on purpose: the drawn cases run after the hand-written ones.)"""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def r3():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import rend3_amd
    return rend3_amd


@pytest.mark.parametrize("mutate,first,count", [(False, 1000, 100), (True, 5000, 60), (True, 40000, 30)])
def test_randomised_cases_bit_exact(r3, mutate, first, count):
    import fuzz_parity as F
    failed = []
    with F.oracle_threads(32):  # small scenes: see fuzz_parity.oracle_threads
        for seed in range(first, first + count):
            c = F.draw_case(seed)
            try:
                (F.run_mutating_case if mutate else F.run_case)(r3, c)
            except AssertionError as e:
                failed.append((seed, str(e)[:300]))
    assert not failed, f"{len(failed)} of {count} cases differ from the oracle (python tools/fuzz_debug.py SEED{' --mutate' if mutate else ''}): {failed[:5]}"
