"""CPU: the texture-interpolation loss oracle against values produced by the real reference methods."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import texture_loss_ref as R  # noqa: E402


def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "texture_loss_golden.npz"))


@pytest.mark.parametrize("tag", ["same", "big"])
def test_oracle_reproduces_reference_values(tag):
    g = gold()
    tex, msk = torch.from_numpy(g["textures"]), torch.from_numpy(g["tx_masks"])
    face = torch.from_numpy(g[f"face_{tag}"])
    assert R.pairwise_texture_loss(face, tex[0], tex[1]).item() == pytest.approx(float(g[f"pair01_{tag}"]), rel=1e-6)
    loop = R.texture_pairs_loss(face, tex, msk, g[f"loop_pairs_{tag}"])
    assert loop.item() == pytest.approx(float(g[f"loop_{tag}"]), rel=1e-6)
    assert len(R.all_pairs(6)) == 10


@pytest.mark.reference
def test_oracle_equals_imported_reference_losses():
    """R1 and the pairwise texture loss of the oracle vs the REAL loss_functions.losses (imported in place)."""
    import types
    from oracle import reference_import as ri
    from oracle import stylegan2_ref as S
    if not ri.available():
        pytest.skip("/root/reference not mounted")
    L = ri.reference_losses()
    g = gold()
    tex = torch.from_numpy(g["textures"])
    me = types.SimpleNamespace(face_region_only_mask=torch.from_numpy(g["face_big"]))
    assert torch.equal(L.InterpolatedTextureLoss.pairwise_texture_loss(me, tex[2], tex[3]),
                       R.pairwise_texture_loss(torch.from_numpy(g["face_big"]), tex[2], tex[3]))
    x = torch.randn(3, 3, 8, 8, requires_grad=True)
    w = torch.randn(3, 8, 8)
    out = (x * w).sum((1, 2, 3)).pow(2)[:, None]
    ref = L.grad_penalty_loss([x], out, step=None)
    got = S.grad_penalty_loss([x], out)
    assert torch.allclose(ref, got, rtol=1e-6, atol=0)
