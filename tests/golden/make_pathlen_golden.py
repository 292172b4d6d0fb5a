"""Generates tests/golden/pathlen_golden.npz with the REAL reference PathLengthRegularizor (loss_functions/losses.py:
102-124) applied to a small stand-in generator (any callable with the generator's keyword signature works: the class only
needs `generator(input=style, noise=None, step=, alpha=, input_indices=)[0]` to be differentiable w.r.t. style).
The random draws (style, pl_noise) are recorded so the HIP-side class can be fed the very same numbers.
Run (build container only): python tests/golden/make_pathlen_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import reference_import as ri  # noqa: E402

B, R = 3, 8


def standin_weights():
    g = torch.Generator().manual_seed(31)
    return torch.randn(159, 3 * R * R, generator=g) / 12, torch.randn(3 * R * R, generator=g)


def standin_generator(A, bias):
    def gen(input, noise=None, step=0, alpha=1, input_indices=None):
        return [torch.tanh(input @ A + bias).view(input.shape[0], 3, R, R) * (1 + 0.1 * input_indices.float().view(-1, 1, 1, 1))]
    return gen


def main():
    L = ri.reference_losses()
    A, bias = standin_weights()
    draws = []
    real_randn = torch.randn

    def logging_randn(*a, **k):
        t = real_randn(*a, **k)
        draws.append(t.detach().clone().numpy())
        return t

    torch.manual_seed(11)
    torch.randn = logging_randn
    try:
        reg = L.PathLengthRegularizor()
        idx = torch.tensor([0, 1, 2])
        p1 = reg.path_length_reg(standin_generator(A, bias), 1, 1.0, idx)
        m1 = float(reg.pl_moving_mean)
        p2 = reg.path_length_reg(standin_generator(A, bias), 1, 1.0, idx)
        m2 = float(reg.pl_moving_mean)
    finally:
        torch.randn = real_randn
    out = {"penalty": np.array([p1.item(), p2.item()], np.float64), "moving_mean": np.array([m1, m2], np.float64)}
    for i, d in enumerate(draws):
        out[f"draw{i}"] = d
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pathlen_golden.npz"), **out)
    print(out["penalty"], out["moving_mean"], [d.shape for d in draws])


if __name__ == "__main__":
    main()
