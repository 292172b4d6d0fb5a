"""bench.py host logic that needs no GPU: where the R1 iterations of a run fall (timed region / warm-up)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _r1_steps(steps, warmup, r1_every):
    it = bench.r1_start_iteration(steps, warmup, r1_every)
    warm = [k for k in range(warmup) if r1_every and (it + k + 1) % r1_every == 0]
    timed = [k for k in range(steps) if r1_every and (it + warmup + k + 1) % r1_every == 0]
    return warm, timed


@pytest.mark.parametrize("steps,warmup,r1_every", [(20, 5, 16), (16, 2, 16), (32, 5, 16), (40, 5, 16), (20, 5, 4), (12, 3, 16), (20, 0, 16),
                                                   (5, 2, 16), (12, 3, 100000), (1, 0, 16), (16, 2, 0)])
def test_r1_share_of_the_timed_region(steps, warmup, r1_every):
    """The timed steps contain round(steps / r1_every) R1 iterations whatever the warm-up does (the reference runs R1 every 16th step)."""
    warm, timed = _r1_steps(steps, warmup, r1_every)
    assert len(timed) == (int(round(steps / r1_every)) if r1_every else 0)
    assert all(b - a == r1_every for a, b in zip(timed, timed[1:]))


@pytest.mark.parametrize("steps,warmup,r1_every", [(20, 5, 16), (16, 2, 16), (32, 5, 16), (40, 5, 16)])
def test_first_r1_iteration_is_a_warmup_step(steps, warmup, r1_every):
    """The driver's and the default command lines: the last warm-up step is an R1 iteration, so that the timed one is not the first of the
    process (buffer allocation, kernel loading)."""
    warm, timed = _r1_steps(steps, warmup, r1_every)
    assert warm == [warmup - 1] and timed and timed[0] == r1_every - 1
