"""Pins oracle/texture_ref.py to outputs of the REAL FlameTextureSpace.compute_texture_map (tests/golden/texture_golden.npz)."""
import os

import numpy as np
import torch

from oracle import texture_ref as TR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_fixture():
    g = np.load(os.path.join(GOLD, "texture_golden.npz"))
    mg = np.load(os.path.join(GOLD, "mesh_golden.npz"))
    T = 256
    ys, xs = np.meshgrid(np.arange(T), np.arange(T), indexing="ij")
    td = {"x_coords": xs.reshape(-1), "y_coords": ys.reshape(-1), "valid_pixel_ids": g["valid_pixel_ids"],
          "valid_pixel_3d_faces": g["valid_pixel_3d_faces"], "valid_pixel_b_coords": g["valid_pixel_b_coords"]}
    return g, td, torch.from_numpy(mg["vertices"]), torch.from_numpy(mg["normals"])


def test_texture_map_oracle_matches_reference_golden():
    g, td, verts, normals = load_fixture()
    img = torch.from_numpy(g["img"]).requires_grad_(True)
    tex, mask = TR.compute_texture_map(td, img, verts, normals, torch.from_numpy(g["cam"]))
    assert np.abs(tex.detach().numpy() - g["tex"]).max() < 1e-6
    assert np.array_equal(mask.numpy(), g["mask"])
    (tex * torch.linspace(-1, 1, tex.numel()).view_as(tex)).sum().backward()
    assert np.abs(img.grad.numpy() - g["grad_img"]).max() <= 1e-4 * np.abs(g["grad_img"]).max()
