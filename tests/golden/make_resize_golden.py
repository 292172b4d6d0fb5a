"""Generates tests/golden/resize_golden.npz with the REAL reference function dataset_loaders.fast_image_reshape
(imported in place from /root/reference through oracle/reference_import.py; build container only).
Run: python tests/golden/make_resize_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import reference_import as ri  # noqa: E402

CASES = [  # (name, B, C, H, W, height_out, width_out, mode, non_diff_allowed)
    ("bicubic_down", 2, 3, 24, 24, 16, 16, "bicubic", False),
    ("bicubic_up", 2, 3, 9, 12, 20, 15, "bicubic", False),
    ("bicubic_clamped", 1, 3, 10, 10, 17, 17, "bicubic", True),
    ("bilinear_down", 2, 6, 32, 32, 8, 8, "bilinear", False),
    ("bilinear_odd", 1, 3, 7, 11, 13, 5, "bilinear", False),
]


def main():
    f = ri.reference_fast_image_reshape()
    out = {}
    for i, (name, B, C, H, W, ho, wo, mode, clamp) in enumerate(CASES):
        x = torch.rand(B, C, H, W, generator=torch.Generator().manual_seed(100 + i)) * 2 - 1
        y = f(x, ho, wo, non_diff_allowed=clamp, mode=mode)
        out[name + "_x"] = x.numpy()
        out[name + "_y"] = y.numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
