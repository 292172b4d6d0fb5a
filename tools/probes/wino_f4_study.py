"""Winograd F(4x4,3x3) against F(2x2,3x3) and the direct convolution in fp32 arithmetic (CPU, numpy / torch; review item 3 of round 4:
"re-open the Winograd tile size with numbers").  F(4x4,3x3): 36 multiplies per 16 outputs (2.25 per output; F(2x2): 4; direct: 9),
V is 36/16 = 2.25 x the input instead of 4 x.  Points {0, +-1, +-2, inf} (Lavin & Gray) and the better-conditioned
{0, +-1, +-1/2, inf}; transforms and the position GEMMs in fp32 (accumulation in chunks of 16 like the MFMA), reference in fp64.

  (a) per layer: the six stride-1 3x3 headline shapes' channel counts (128, 256, 512; randn and post-leaky-ReLU activations),
      error = max |y - y64| / max |y64|;
  (b) whole model: the oracle generator (oracle/stylegan2_ref.py) with every Winograd-eligible layer (stride-1 3x3 modulated convs
      with >= 32 input and >= 48 output channels) computed by F(4x4,3x3) resp. F(2x2,3x3) in fp32, L_inf of the image against the
      oracle's own direct fp32 convolutions.
Usage: python tools/probes/wino_f4_study.py [--model-res 64]"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")


def f23():
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    return BT, G, AT


def f43(points):
    """Cook-Toom F(4,3) for the given 5 finite points (+ infinity): matrices by solving the defining identities numerically in fp64"""
    if points == "lavin":
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                     np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
        return BT, G, AT
    # points {0, 1, -1, 1/2, -1/2, inf}: derive by the transposition principle from the polynomial-evaluation matrices
    pts = [0.0, 1.0, -1.0, 0.5, -0.5]
    n, r = 4, 3
    a = n + r - 1  # 6
    # evaluation matrices: V_m [a x m] with rows (1, p, p^2, ...) and the infinity row (0, ..., 1)
    def vand(m):
        M = np.zeros((a, m))
        for i, p in enumerate(pts):
            M[i] = [p ** k for k in range(m)]
        M[a - 1, m - 1] = 1.0
        return M
    Vn, Vr, Va = vand(n), vand(r), vand(a)
    # linear convolution s = Va^-1 [(Vn d) * (Vr g)]; correlation (FIR) form by transposition: y = Vn^T [(Vr g) * (Va^-T x)]
    AT = Vn.T
    G = Vr
    BT = np.linalg.inv(Va).T
    return BT, G, AT


def wino_conv(x, w, mats, m, chunk=16):
    """x [C, H, W] fp32, w [O, C, 3, 3] fp32, stride 1 pad 1; tiles of m x m outputs; everything fp32, K accumulated in chunks"""
    BT, G, AT = (t.astype(np.float32) for t in mats)
    C, H, W = x.shape
    O = w.shape[0]
    a = m + 2
    assert H % m == 0 and W % m == 0
    xp = np.zeros((C, H + 2, W + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    th, tw = H // m, W // m
    # gather tiles [C, th, tw, a, a]
    d = np.empty((C, th, tw, a, a), np.float32)
    for i in range(th):
        for j in range(tw):
            d[:, i, j] = xp[:, i * m:i * m + a, j * m:j * m + a]
    V = np.einsum("pa,ctuab,qb->ctupq", BT, d, BT, optimize=True).astype(np.float32)
    U = np.einsum("pa,ocab,qb->ocpq", G, w, G, optimize=True).astype(np.float32)
    M = np.zeros((O, th, tw, a, a), np.float32)
    for c0 in range(0, C, chunk):
        M += np.einsum("ocpq,ctupq->otupq", U[:, c0:c0 + chunk], V[c0:c0 + chunk], optimize=True).astype(np.float32)
    Y = np.einsum("ip,otupq,jq->otuij", AT, M, AT, optimize=True).astype(np.float32)
    return Y.transpose(0, 1, 3, 2, 4).reshape(O, H, W)


def direct32(x, w, chunk=16):
    xt, wt = torch.from_numpy(x)[None], torch.from_numpy(w)
    acc = torch.zeros((1, w.shape[0]) + x.shape[1:], dtype=torch.float32)
    for c0 in range(0, x.shape[0], chunk):
        acc += F.conv2d(xt[:, c0:c0 + chunk], wt[:, c0:c0 + chunk], padding=1)
    return acc[0].numpy()


def per_layer():
    rng = np.random.RandomState(0)
    print("(a) per layer, error = max |y - y64| / max |y64| (fp32 transforms, fp32 position GEMMs in chunks of 16):")
    print(f"{'C -> O, data':34s} {'direct fp32':>12s} {'F(2x2)':>10s} {'F(4x4) Lavin':>13s} {'F(4x4) +-1/2':>13s}")
    for C in (128, 256, 512):
        for kind in ("randn", "lrelu"):
            H = 16
            x = rng.randn(C, H, H).astype(np.float32)
            if kind == "lrelu":
                x = (np.where(x > 0, x, 0.2 * x) * 2 ** 0.5 + 0.3).astype(np.float32)
            w = (rng.randn(C, C, 3, 3) / (C * 9) ** 0.5).astype(np.float32)
            ref = F.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), padding=1)[0].numpy()
            e = lambda y: float(np.abs(y - ref).max() / np.abs(ref).max())
            print(f"{C:4d} -> {C:<4d} {kind:20s} {e(direct32(x, w)):12.2e} {e(wino_conv(x, w, f23(), 2)):10.2e} "
                  f"{e(wino_conv(x, w, f43('lavin'), 4)):13.2e} {e(wino_conv(x, w, f43('half'), 4)):13.2e}", flush=True)


def whole_model(res):
    from oracle import stylegan2_ref as R
    import contextlib
    import io
    from gif_amd.generator import StyledGenerator
    step = {32: 3, 64: 4, 128: 5, 256: 6}[res]
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        g = StyledGenerator(embedding_vocab_size=16, rendered_flame_ascondition=True, normal_maps_as_cond=True)
    sd = R.seeded_state_dict(g.state_dict(), 3)
    cond = torch.rand(2, 6, res, res) * 2 - 1
    idx = torch.tensor([1, 5])
    with torch.no_grad():
        ref = R.generator_forward(sd, cond, step, idx)
    orig = F.conv2d
    out = {}
    for name, mats, m in (("F(2x2,3x3)", f23(), 2), ("F(4x4,3x3) Lavin", f43("lavin"), 4), ("F(4x4,3x3) +-1/2", f43("half"), 4)):
        n_layers = [0]

        def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
            k = w.shape[-1]
            cin = w.shape[1]
            cout = w.shape[0] // groups
            H, W = x.shape[-2:]
            if (k == 3 and stride == 1 and padding == 1 and cin >= 32 and cout >= 48 and H % m == 0 and W % m == 0 and H >= 8):
                n_layers[0] += 1
                ys = []
                for gi in range(groups):  # the oracle's per-sample weights: one group per sample
                    xg = x[0, gi * cin:(gi + 1) * cin].numpy().astype(np.float32)
                    wg = w[gi * cout:(gi + 1) * cout].numpy().astype(np.float32)
                    ys.append(torch.from_numpy(wino_conv(xg, wg, mats, m)))
                y = torch.cat(ys, 0)[None]
                return y if bias is None else y + bias[None, :, None, None]
            return orig(x, w, bias, stride, padding, dilation, groups)
        F.conv2d = conv2d
        try:
            with torch.no_grad():
                got = R.generator_forward(sd, cond, step, idx)
        finally:
            F.conv2d = orig
        out[name] = (float((got - ref).abs().max()), n_layers[0])
    print(f"\n(b) whole generator at {res}x{res}, batch 2 (image max {float(ref.abs().max()):.2f}); L_inf of the image vs the oracle's direct fp32 "
          "convolutions, eligible layers computed by:")
    for k, (e, n) in out.items():
        print(f"   {k:20s} L_inf {e:.2e}   ({n} layer evaluations)")


if __name__ == "__main__":
    per_layer()
    res = int(sys.argv[sys.argv.index("--model-res") + 1]) if "--model-res" in sys.argv else 64
    whole_model(res)
