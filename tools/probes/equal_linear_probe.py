"""Diagnostic (GPU box): EqualLinear first-order gradients vs the oracle, with intermediates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from gif_amd import layers as L, ops
from oracle import stylegan2_ref as R
torch.manual_seed(11)
m = L.EqualLinear(512, 512, lr_mul=0.01, activation='fused_lrelu').cuda()
with torch.no_grad():
    m.bias.add_(torch.randn_like(m.bias))
x = torch.randn(7, 512)
xr = x.clone().requires_grad_(True); wr = m.weight.detach().cpu().clone().requires_grad_(True); br = m.bias.detach().cpu().clone().requires_grad_(True)
ref = R.equal_linear(xr, wr, br, lr_mul=0.01, activation=True)
xd = x.cuda().requires_grad_(True)
got = m(xd)
gy = torch.randn(ref.shape)
for cg in (False, True):
    gr = torch.autograd.grad(ref, [xr, wr, br], gy, create_graph=cg, retain_graph=True)
    gd = torch.autograd.grad(got, [xd, m.weight, m.bias], gy.cuda(), create_graph=cg, retain_graph=True)
    for a, b, n in zip(gd, gr, "xwb"):
        e = (a.detach().cpu() - b.detach()).abs()
        bad = (e > 1e-3 * b.abs().max()).nonzero()
        print("create_graph", cg, n, f"max err {e.max().item():.3e} ref max {b.abs().max().item():.3e} bad {len(bad)} of {e.numel()}",
              "rows", sorted(set(bad[:, 0].tolist()))[:10] if len(bad) else [], "cols", sorted(set(bad[:, -1].tolist()))[:10] if len(bad) else [])
# pieces
y = got.detach()
mask = torch.where(y > 0, 1.0, 0.2)
gpre = gy.cuda() * mask
gx_manual = (gpre @ m.weight.detach()) * m.scale
print("manual gx vs oracle", ((gx_manual.cpu() - gr[0].detach()).abs().max() / gr[0].abs().max()).item())
print("ops.linear_nn vs manual", ((ops.linear_nn(gpre.contiguous(), m.weight.detach(), m.scale) - gx_manual).abs().max() / gx_manual.abs().max()).item())
flips = ((y.cpu() > 0) != (ref.detach() > 0)).sum().item()
print("sign flips fwd", flips, "min |ref|", ref.abs().min().item())
