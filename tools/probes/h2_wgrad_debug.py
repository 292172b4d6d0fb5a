"""Where is the f16x2 128x128 weight-gradient kernel wrong?  (debug probe)"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402

ops.WINOGRAD = False
dev = "cuda"
B, ci, co, k, h = 4, 128, 128, 3, 64
spec = ops.ConvSpec(k, k, 1, 1)


def run(x, gy, tag):
    wd = torch.zeros(co, ci, k, k, device=dev, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x.double(), wd, padding=1), wd, gy.double())
    ops.set_fp32_mfma_mode("f16x2")
    got = ops.conv_wgrad(gy, x, spec, co, ci).double()
    ops.set_fp32_mfma_mode("bf16x3")
    got3 = ops.conv_wgrad(gy, x, spec, co, ci).double()
    e = (got - ref).abs() / ref.abs().max()
    print(tag, "max err h2", float(e.max()), "x3", float(((got3 - ref).abs() / ref.abs().max()).max()))
    bad = e > 1e-4
    print("   bad entries", int(bad.sum()), "of", bad.numel(), "| per tap", bad.sum(dim=(0, 1)).flatten().tolist())
    print("   bad rows (cout) count", int(bad.any(dim=(1, 2, 3)).sum()), "bad cols (cin) count", int(bad.any(dim=(0, 2, 3)).sum()))
    rows = bad.any(dim=(1, 2, 3)).nonzero().flatten().tolist()
    cols = bad.any(dim=(0, 2, 3)).nonzero().flatten().tolist()
    print("   rows", rows[:40], "\n   cols", cols[:40])
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        print("   first bad", idx, "got", float(got[tuple(idx)]), "ref", float(ref[tuple(idx)]), "ratio", float(got[tuple(idx)] / ref[tuple(idx)]))


torch.manual_seed(0)
ones_x = torch.ones(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
ones_g = torch.ones(B, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
rx = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
rg = torch.randn(B, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
run(ones_x, ones_g, "ones x ones")
run(rx, ones_g, "randn x, ones gy")
run(ones_x, rg, "ones x, randn gy")
run(rx, rg, "randn both")
run(rx.abs() + 1, rg.abs() + 1, "|randn|+1 both (no rescale after the first group: max < 4x min)")

# pattern of the bad entries for ones x ones
wd = torch.zeros(co, ci, k, k, device=dev, dtype=torch.float64, requires_grad=True)
(ref,) = torch.autograd.grad(F.conv2d(ones_x.double(), wd, padding=1), wd, ones_g.double())
ops.set_fp32_mfma_mode("f16x2")
got = ops.conv_wgrad(ones_g, ones_x, spec, co, ci).double()
diff = (got - ref)
for t in range(9):
    d = diff[:, :, t // 3, t % 3]
    blocks = [[int((d[a * 32:(a + 1) * 32, b * 32:(b + 1) * 32] != 0).sum()) for b in range(4)] for a in range(4)]
    vals = sorted(set(d.flatten().tolist()))
    print("tap", t, "bad per 32x32 block (rows = cout blocks)", blocks, "values", vals[:8])
d = diff[:, :, 0, 1]
print("tap 1 bad rows%32 histogram", [(r, int((d[r::32] != 0).sum())) for r in range(32)])
print("tap 1 bad cols%32 histogram", [(c, int((d[:, c::32] != 0).sum())) for c in range(32)])
