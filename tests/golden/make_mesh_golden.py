"""Generates tests/golden/mesh_golden.npz with the REAL reference helpers (model/mesh_and_3d_helpers.py imported in
place; build container only):  python tests/golden/make_mesh_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from model import mesh_and_3d_helpers as M  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "body_mesh.npz"))
rng = np.random.RandomState(3)
v = np.stack([g["vertices"], g["vertices"] * 0.7 + rng.normal(0, 0.01, g["vertices"].shape).astype(np.float32)]).astype(np.float32)
f = np.repeat(g["faces"][None], 2, 0)
cam = np.array([[1.3, 0.1, -0.2], [0.8, -0.05, 0.3]], np.float32)
n = M.vertex_normals(torch.from_numpy(v), torch.from_numpy(f).long()).numpy()
p = M.batch_orth_proj(torch.from_numpy(v), torch.from_numpy(cam)).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mesh_golden.npz"), vertices=v, cam=cam, normals=n, proj=p)
print("wrote mesh_golden.npz", n.shape, p.shape)
