"""f16 activations: per-parameter gradient differences between the gradient-epilogue fusions on / off, next to the f16-vs-fp32
noise floor of the same tensors (is a 9 % max-abs difference of one tensor rounding noise or a defect?).  GPU probe."""
import contextlib
import io
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402
from gif_amd.discriminator import Discriminator  # noqa: E402
from gif_amd.generator import StyledGenerator  # noqa: E402

res, step, B = 64, 4, 8
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    G = StyledGenerator(embedding_vocab_size=16, rendered_flame_ascondition=True, normal_maps_as_cond=True).cuda()
    D = Discriminator(size=res, num_color_chnls=9).cuda()
cond = (torch.rand(B, 6, res, res) * 2 - 1).cuda()
real = (torch.rand(B, 3, res, res) * 2 - 1).cuda()
idx = torch.randint(0, 16, (B,)).cuda()
gnames = [n for n, p in G.named_parameters() if not any(f"progression.{i}." in n or f"to_rgb.{i}." in n for i in range(step + 1, 9))]
gp = [dict(G.named_parameters())[n] for n in gnames]
dnames = [n for n, _ in D.named_parameters()]
dp = list(D.parameters())


def run(fused, dt):
    G.set_activation_dtype(dt), D.set_activation_dtype(dt)
    ops.FUSE_GRAD = fused
    fake = G(cond, None, step=step, alpha=1, input_indices=idx)
    lg = F.softplus(-D(fake, condition=cond, step=step, alpha=1)[0]).mean()
    gg = torch.autograd.grad(lg, gp, allow_unused=True)
    ld = F.softplus(-D([real], condition=cond, step=step, alpha=1)[0]).mean() + F.softplus(D([fake[0].detach()], condition=cond, step=step, alpha=1)[0]).mean()
    gd = torch.autograd.grad(ld, dp)
    return list(gg) + list(gd)


ref = run(False, torch.float32)
u16 = run(False, torch.float16)
f16 = run(True, torch.float16)
rows = []
for n, r, u, f in zip(["G." + n for n in gnames] + ["D." + n for n in dnames], ref, u16, f16):
    if r is None:
        continue
    m = r.abs().max().item() + 1e-30
    rows.append((((f - u).abs().max() / m).item(), ((u - r).abs().max() / m).item(), ((f - r).abs().max() / m).item(), n, tuple(r.shape)))
rows.sort(reverse=True)
print("fused-vs-unfused | unfused-vs-fp32 | fused-vs-fp32 (max-abs / fp32 max) | parameter")
for a, b, c, n, sh in rows[:25]:
    print(f"{a:9.2e} {b:9.2e} {c:9.2e}  {n} {sh}")
print("median fused-vs-unfused", sorted(r[0] for r in rows)[len(rows) // 2], "median unfused-vs-fp32", sorted(r[1] for r in rows)[len(rows) // 2])
