"""Pins the rasteriser oracle (oracle/rasterize_ref.c) to the reference's golden OBJ fixtures
(my_utils/standard_rasterize_cuda/data/obj/body_vis.obj, body_vis_z.obj via tests/golden/body_mesh.npz,
demo_vert_visibility.py:12-22: verts*0.8, h=w=512)."""
import os

import numpy as np

from oracle import rasterize_oracle as ro

GOLD = os.path.join(os.path.dirname(__file__), "golden", "body_mesh.npz")


def _mesh():
    g = np.load(GOLD)
    v = (g["vertices"] * np.float32(0.8))[None]
    f = g["faces"][None]
    return g, v, f


def test_get_visibility_matches_reference_golden():
    g, v, f = _mesh()
    vis, (depth, tri, bary) = ro.get_visibility(v, f, 512, 512)
    assert np.array_equal(vis[0].astype(np.uint8), g["vis"])
    assert int((tri >= 0).sum()) == 29610  # covered pixels


def test_get_visibility_z_matches_reference_golden():
    g, v, f = _mesh()
    vis, _ = ro.get_visibility_z(v, f, 512, 512)
    assert np.array_equal(vis[0].astype(np.uint8), g["vis_z"])


def test_bary_and_colors_consistent():
    g, v, f = _mesh()
    vi = ro.to_image_space(v, 128, 128)
    fv = ro.face_vertices(vi, f)
    d1, t1, b1 = ro.new_buffers(1, 128, 128)
    ro.standard_rasterize(fv, d1, t1, b1, 128, 128)
    # colours = vertex positions -> image must equal bary-weighted positions of the winning face
    d2, t2, img = ro.new_buffers(1, 128, 128)
    ro.standard_rasterize_colors(fv, fv.copy(), d2, t2, img, 128, 128)
    assert np.array_equal(t1, t2) and np.array_equal(d1, d2)
    m = t1[0] >= 0
    s = b1[0][m].sum(-1)
    assert np.allclose(s, 1.0, atol=1e-5)
    win = fv[0][t1[0][m]]  # [n,3,3]
    bw = b1[0][m]
    exp = bw[:, 0:1] * win[:, 0] + bw[:, 1:2] * win[:, 1] + bw[:, 2:3] * win[:, 2]
    assert np.allclose(img[0][m], exp, rtol=1e-5, atol=1e-4)
    # uncovered pixels untouched
    assert (d1[0][~m] == np.float32(1e6)).all() and (b1[0][~m] == 0).all()


def test_empty_and_backfacing():
    d, t, b = ro.new_buffers(2, 8, 8)
    fv = np.zeros((2, 0, 3, 3), np.float32)
    ro.standard_rasterize(fv, d, t, b, 8, 8)
    assert (t == -1).all()
    # one CCW-in-image (back-facing by the reference's test) and one front-facing triangle
    tri_front = np.array([[[1, 1, 2], [1, 6, 2], [6, 1, 2]]], np.float32)
    tri_back = tri_front[:, ::-1].copy()
    for fv1, covered in ((tri_front, True), (tri_back, False)):
        d, t, b = ro.new_buffers(1, 8, 8)
        ro.standard_rasterize(np.ascontiguousarray(fv1[None]), d, t, b, 8, 8)
        assert bool((t >= 0).any()) == covered
