"""ORACLE — TEST INFRASTRUCTURE ONLY.  torch-CPU restatement of FlameTextureSpace.compute_texture_map
(model/stg2_generator.py:378-421).  Pinned against the real reference method (run on a synthetic texture_data fixture,
tests/golden/make_texture_golden.py) — `tests/test_oracle_texture.py`."""
import numpy as np
import torch
import torch.nn.functional as F


def synthetic_texture_data(rng, n_faces, T=256, n_valid=9000):
    """A stand-in for the licensed FLAME texture data file: random valid texels with a face + barycentrics each."""
    ys, xs = np.meshgrid(np.arange(T), np.arange(T), indexing="ij")
    x_coords, y_coords = xs.reshape(-1), ys.reshape(-1)
    valid = np.sort(rng.choice(T * T, n_valid, replace=False))
    bc = rng.dirichlet([1, 1, 1], n_valid).astype(np.float32)
    return {"x_coords": x_coords, "y_coords": y_coords, "valid_pixel_ids": valid,
            "valid_pixel_3d_faces_idx": rng.randint(0, n_faces, n_valid), "valid_pixel_b_coords": bc}


def compute_texture_map(texture_data, source_img, verts, vertex_normals, cam, T=256):
    x = np.asarray(texture_data["x_coords"]).astype("int")
    y = np.asarray(texture_data["y_coords"]).astype("int")
    ids = np.asarray(texture_data["valid_pixel_ids"]).astype("int")
    f = torch.from_numpy(np.asarray(texture_data["valid_pixel_3d_faces"]).astype("int64"))
    bc = torch.from_numpy(np.asarray(texture_data["valid_pixel_b_coords"]).astype("float32"))
    p3 = sum(verts[:, f[:, k], :] * bc[:, k][None, :, None] for k in range(3))               # :386-389
    camv = cam.clone().view(-1, 1, 3)
    proj = (camv[:, :, 0:1] * torch.cat([p3[:, :, :2] + camv[:, :, 1:], p3[:, :, 2:]], 2))[:, :, :2].clone()  # :399
    proj[:, :, 1] *= -1                                                                       # :400
    grid = torch.zeros((source_img.shape[0], T, T, 2), dtype=torch.float32)                    # :402
    grid[:, y[ids], x[ids], :] = proj
    tex = F.grid_sample(source_img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)  # :406 (torch 1.7 default)
    n3 = sum(vertex_normals[:, f[:, k], :] * bc[:, k][:, None] for k in range(3))             # :409-412
    mask = torch.zeros((source_img.shape[0], 1, T, T), dtype=torch.bool)
    mask[:, :, y[ids], x[ids]] = (n3[:, :, -1:] < 0).transpose(1, 2)                          # :413-417
    return tex, mask
