"""Diagnostic for csrc/linear.hip (run on the GPU box): each product against torch, with the error pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gif_amd import ops
g = torch.Generator().manual_seed(1)
for M, N, K in [(7, 512, 512), (32, 512, 512), (32, 64, 64), (4, 512, 8192)]:
    a = torch.randn(M, K, generator=g); b = torch.randn(N, K, generator=g); gy = torch.randn(M, N, generator=g)
    ad, bd, gd = a.cuda(), b.cuda(), gy.cuda()
    for name, got, ref in (("nt", ops.linear_nt(ad, bd), a @ b.t()), ("nn", ops.linear_nn(gd, bd), gy @ b), ("tn", ops.linear_tn(gd, ad), gy.t() @ a)):
        e = (got.cpu() - ref).abs()
        bad = (e > 1e-3 * ref.abs().max()).nonzero()
        print(f"{M}x{N}x{K} {name}: max err {e.max().item():.3e} (ref max {ref.abs().max().item():.2f}), bad {len(bad)} of {e.numel()}",
              "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:12])
