"""Generates tests/golden/texture_loss_golden.npz by calling the REAL reference methods
InterpolatedTextureLoss.pairwise_texture_loss / tex_sp_intrp_loss (loss_functions/losses.py:147-176) — unbound, on a
stand-in `self` that carries the attributes they read (the class constructor needs licensed FLAME files).
Run (build container only): python tests/golden/make_texture_loss_golden.py"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import reference_import as ri  # noqa: E402

N, T = 6, 32  # max_images_in_batch = 6 -> max_num = 5 textures


def inputs():
    g = torch.Generator().manual_seed(77)
    textures = torch.rand(N - 1, 3, T, T, generator=g) * 2 - 1
    tx_masks = torch.rand(N - 1, 1, T, T, generator=g) > 0.3
    face_same = (torch.rand(1, 1, T, T, generator=g) > 0.4).float()
    face_big = torch.rand(1, 1, 48, 48, generator=g)
    return textures, tx_masks, face_same, face_big


def main():
    L = ri.reference_losses().InterpolatedTextureLoss
    textures, tx_masks, face_same, face_big = inputs()
    out = {"textures": textures.numpy(), "tx_masks": tx_masks.numpy(), "face_same": face_same.numpy(), "face_big": face_big.numpy()}
    for tag, face in (("same", face_same), ("big", face_big)):
        me = types.SimpleNamespace(face_region_only_mask=face.clone())
        me.pairwise_texture_loss = lambda tx1, tx2, me=me: L.pairwise_texture_loss(me, tx1, tx2)
        out[f"pair01_{tag}"] = me.pairwise_texture_loss(textures[0], textures[1]).numpy()
        # the whole tex_sp_intrp_loss with a stand-in get_image_and_textures (the FLAME part) and a seeded pair draw
        me.max_num = N - 1
        me.pairs = np.array([(i, j) for i in range(me.max_num) for j in range(i + 1, me.max_num)])
        me.get_image_and_textures = lambda *a, **k: (textures, tx_masks, None)
        np.random.seed(5)
        out[f"loop_{tag}"] = L.tex_sp_intrp_loss(me, None, None, 0, 1.0, 10, True, False, True).numpy()
        np.random.seed(5)
        out[f"loop_pairs_{tag}"] = me.pairs[np.random.choice(len(me.pairs), me.max_num, replace=False)]
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "texture_loss_golden.npz"), **out)
    print({k: (v.shape, float(v) if v.ndim == 0 else None) for k, v in out.items()})


if __name__ == "__main__":
    main()
