"""CPU: the resize oracle (oracle/resize_ref.py) against golden vectors produced by the real reference function
dataset_loaders.fast_image_reshape, and its explicit tap-by-tap form against ATen."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden.make_resize_golden import CASES  # noqa: E402
from oracle import resize_ref as R  # noqa: E402


def _gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "resize_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_reference_goldens(case):
    name, B, C, H, W, ho, wo, mode, clamp = case
    g = _gold()
    x = torch.from_numpy(g[name + "_x"])
    y = R.fast_image_reshape(x, ho, wo, non_diff_allowed=clamp, mode=mode)
    assert y.shape == (B, C, wo, ho)  # the reference passes (width_out, height_out) as (rows, cols)
    assert np.array_equal(y.numpy(), g[name + "_y"]), "same ATen kernels, same inputs: bit-identical"
    e = R.resize_explicit(g[name + "_x"], (wo, ho), mode)
    if clamp:
        e = np.clip(e, g[name + "_x"].min(), g[name + "_x"].max())
    assert np.abs(e - g[name + "_y"]).max() < 2e-6, "explicit tap form vs the reference output"


def test_explicit_taps_partition_unity_and_identity():
    for mode in ("bilinear", "bicubic"):
        for n_in, n_out in ((7, 7), (5, 12), (33, 8)):
            idx, w = R._taps(n_out, n_in, mode)
            assert np.allclose(w.sum(1), 1, atol=1e-6) and idx.min() >= 0 and idx.max() <= n_in - 1
        x = np.random.RandomState(0).rand(1, 2, 6, 9).astype(np.float32)
        assert np.abs(R.resize_explicit(x, (6, 9), mode) - x).max() < 1e-6  # same size = identity


@pytest.mark.reference
def test_oracle_equals_imported_reference():
    from oracle import reference_import as ri
    if not ri.available():
        pytest.skip("/root/reference not mounted")
    f = ri.reference_fast_image_reshape()
    x = torch.rand(2, 3, 13, 10, generator=torch.Generator().manual_seed(5))
    for mode in ("bicubic", "bilinear"):
        for clamp in (False, True):
            assert torch.equal(f(x, 21, 6, non_diff_allowed=clamp, mode=mode),
                               R.fast_image_reshape(x, 21, 6, non_diff_allowed=clamp, mode=mode))
