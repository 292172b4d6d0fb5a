"""Pins oracle/mesh_ref.py (vertex_normals, batch_orth_proj) to golden outputs of the real reference helpers
(model/mesh_and_3d_helpers.py:5-50, tests/golden/make_mesh_golden.py)."""
import os

import numpy as np

from oracle import mesh_ref as MR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_vertex_normals_and_orth_proj_match_reference_golden():
    g = np.load(os.path.join(GOLD, "mesh_golden.npz"))
    faces = np.load(os.path.join(GOLD, "body_mesh.npz"))["faces"]
    n = MR.vertex_normals(g["vertices"], faces)
    assert np.abs(n - g["normals"]).max() < 5e-6  # fp32 rounding (FMA contraction differs between torch and numpy)
    assert np.allclose(np.linalg.norm(n, axis=2), 1.0, atol=1e-5)
    p = MR.batch_orth_proj(g["vertices"], g["cam"])
    assert np.array_equal(p, g["proj"])


def test_csr_order_matches_reference_passes():
    """The product's gather order (corner 1, 2, 0; faces ascending) is the reference's index_add_ order."""
    from gif_amd.render import _topology_csr
    faces = np.array([[0, 1, 2], [2, 1, 3], [0, 2, 3]], np.int64)
    ent, counts = _topology_csr(faces)
    assert counts.tolist() == [2, 2, 3, 2]
    # vertex 2: corner-1 pass has none, corner-2 pass: face 0; corner-0 pass: face 1 ; corner 1 of face 2 comes first
    v2 = ent[counts[:2].sum():counts[:3].sum()].tolist()
    assert v2 == [2 * 4 + 1, 0 * 4 + 2, 1 * 4 + 0]
