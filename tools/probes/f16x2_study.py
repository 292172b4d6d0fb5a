import numpy as np
rng = np.random.default_rng(0)
def bf16_trunc_rn(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)
def split_bf16x3(x):
    h = bf16_trunc_rn(x); r = x - h
    m = bf16_trunc_rn(r); r2 = r - m
    l = bf16_trunc_rn(r2)
    return h, m, l
def split_f16x2(x, s):
    xs = (x * s).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    l = (xs - h).astype(np.float16).astype(np.float32)
    return h, l
def acc32(prod):  # fp32 accumulation along axis -1 in chunks of 16 (MFMA-like: sum 16 exactly-ish then add)
    p = prod.astype(np.float64).reshape(prod.shape[0], -1, 16).sum(-1).astype(np.float32)
    out = np.zeros(p.shape[0], np.float32)
    for k in range(p.shape[1]):
        out = (out + p[:, k]).astype(np.float32)
    return out
for K, dist in [(4608, "normal"), (1152, "normal"), (4608, "lognormal"), (576, "sparse")]:
    M = 2048
    if dist == "normal":
        a = rng.standard_normal((M, K)).astype(np.float32); b = rng.standard_normal((M, K)).astype(np.float32)
    elif dist == "lognormal":
        a = (rng.standard_normal((M, K)) * np.exp(3 * rng.standard_normal((M, K)))).astype(np.float32)
        b = rng.standard_normal((M, K)).astype(np.float32) * 0.02
    else:
        a = (rng.standard_normal((M, K)) * (rng.random((M, K)) < 0.05) * 1e-6).astype(np.float32)
        b = rng.standard_normal((M, K)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
    scale = np.abs(a.astype(np.float64) * b.astype(np.float64)).sum(-1)
    # native fp32
    nat = acc32(a.astype(np.float64) * b.astype(np.float64))
    # bf16x3, 6 products
    ah, am, al = split_bf16x3(a); bh, bm, bl = split_bf16x3(b)
    p6 = (ah.astype(np.float64)*bh + ah.astype(np.float64)*bm + am.astype(np.float64)*bh + ah.astype(np.float64)*bl + am.astype(np.float64)*bm + al.astype(np.float64)*bh)
    x3 = acc32(p6)
    # f16x2: per-row power-of-two scale so that row amax -> [2^13, 2^14)
    def rowscale(x):
        am_ = np.abs(x).max(-1, keepdims=True); am_[am_ == 0] = 1
        return np.exp2(13 - np.floor(np.log2(am_))).astype(np.float32)
    sa, sb = rowscale(a), rowscale(b)
    fh, fl = split_f16x2(a, sa); gh, gl = split_f16x2(b, sb)
    p3 = (fh.astype(np.float64)*gh + fh.astype(np.float64)*gl + fl.astype(np.float64)*gh)
    f2 = acc32(p3) / (sa[:, 0] * sb[:, 0])
    # bf16x2: 3 products
    p3b = (ah.astype(np.float64)*bh + ah.astype(np.float64)*bm + am.astype(np.float64)*bh)
    b2 = acc32(p3b)
    def e(x): return np.abs(x - ref).max() / np.abs(ref).max(), np.sqrt(np.mean(((x - ref) / scale) ** 2))
    print(f"K={K} {dist:9s} native {e(nat)[0]:.2e}/{e(nat)[1]:.2e}  bf16x3 {e(x3)[0]:.2e}/{e(x3)[1]:.2e}  f16x2 {e(f2)[0]:.2e}/{e(f2)[1]:.2e}  bf16x2 {e(b2)[0]:.2e}/{e(b2)[1]:.2e}")
