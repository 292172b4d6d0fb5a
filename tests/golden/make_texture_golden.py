"""Generates tests/golden/texture_golden.npz by running the REAL FlameTextureSpace.compute_texture_map
(model/stg2_generator.py:378-421, imported in place with stubs; its FLAME-dependent __init__ is bypassed with
object.__new__) on a synthetic texture_data fixture.  Build container only:  python tests/golden/make_texture_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_import as ri  # noqa: E402
from oracle import texture_ref as TR  # noqa: E402

ri.reference_modules()
from model import stg2_generator as SG  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "body_mesh.npz"))
mg = np.load(os.path.join(ROOT, "tests", "golden", "mesh_golden.npz"))
rng = np.random.RandomState(7)
faces = g["faces"]
td = TR.synthetic_texture_data(rng, len(faces))
td["valid_pixel_3d_faces"] = faces[td.pop("valid_pixel_3d_faces_idx")]
verts = torch.from_numpy(mg["vertices"])          # [2,V,3]
normals = torch.from_numpy(mg["normals"])
cam = torch.tensor([[1.1, 0.02, 0.35], [0.9, -0.1, 0.2]])
img = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1

obj = object.__new__(SG.FlameTextureSpace)
torch.nn.Module.__init__(obj)
obj.texture_data = td
obj.x_coords = td["x_coords"].astype("int")
obj.y_coords = td["y_coords"].astype("int")
obj.valid_pixel_ids = td["valid_pixel_ids"].astype("int")
obj.valid_pixel_3d_faces = torch.from_numpy(td["valid_pixel_3d_faces"].astype("int"))
obj.valid_pixel_b_coords = torch.from_numpy(td["valid_pixel_b_coords"].astype("float32"))
img_r = img.clone().requires_grad_(True)
tex, mask = obj.compute_texture_map(img_r, verts, normals, camera_params=cam)
w = torch.linspace(-1, 1, tex.numel()).view_as(tex)
(tex * w).sum().backward()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "texture_golden.npz"), img=img.numpy(), cam=cam.numpy(),
                    valid_pixel_ids=td["valid_pixel_ids"], valid_pixel_3d_faces=td["valid_pixel_3d_faces"],
                    valid_pixel_b_coords=td["valid_pixel_b_coords"], tex=tex.detach().numpy(), mask=mask.numpy(),
                    grad_img=img_r.grad.numpy())
print("wrote texture_golden.npz", tex.shape, mask.float().mean().item())
