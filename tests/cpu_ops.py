"""TEST INFRASTRUCTURE ONLY — a torch-CPU restatement of the CONTRACT of every gif_amd.ops entry point the model code calls
(what each launch must compute, in terms of ATen ops), installed over gif_amd.ops by the `cpu_ops` fixture.

Purpose: the autograd wiring of gif_amd/functional.py, layers.py, generator.py, discriminator.py, losses.py and
train_step.py (which backward calls which op with which operands, the any-order Function compositions, the recorded
backward of the generator, the data-parallel trainer) is plain Python and can be checked against the oracle in the
`-m "not gpu"` tier and in world_size-2 gloo processes — without a GPU and without touching the kernels.  The kernels
themselves are checked against the same oracle in the `-m gpu` tier.  Nothing under gif_amd/ imports this module: the product
path has no CPU route (gif_amd.ops.nhwc raises on CPU tensors unless a test has patched it)."""
import torch
import torch.nn.functional as F

CL = torch.channels_last


def _cpad(c, dtype=torch.float32):
    q = 8 if dtype == torch.float16 else 4
    return (c + q - 1) // q * q


def nhwc(x):
    assert x.dim() == 4
    return x.contiguous(memory_format=CL)


def _pad_c(x, c):
    return x if x.shape[1] == c else F.pad(x, (0, 0, 0, 0, 0, c - x.shape[1]))


def _epilogue(z, cact, in_scale=None, out_scale=None, bias=None, residual=None, act=False, slope=0.2, gain=2 ** 0.5, fuse=None,
              out_f32=False):
    """fuse: gif_amd.ops.GradFuse — the gradient-producer fusions of gif_conv_epilogue ABI 2 (include/gif_hip.h)."""
    z = _pad_c(z, cact)
    if fuse is not None and fuse.dot_src is not None:
        fuse.dot = (z * fuse.dot_src).sum(dim=(2, 3))  # contraction * dot_src, BEFORE out_scale
    if out_scale is not None:
        z = z * out_scale[:, :, None, None]
    if residual is not None:
        z = z + residual
    if bias is not None:
        z = z + bias[None, :, None, None]
    if act:
        z = gain * F.leaky_relu(z, slope)
    if fuse is not None:
        if fuse.mask_src is not None:
            m = fuse.mask_src
            z = z * fuse.mask_gain * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, fuse.mask_slope))
        if fuse.want_colsum:
            fuse.colsum = z.sum(dim=(0, 2, 3))
    return z.contiguous(memory_format=CL)


def conv_fwd(big, w, spec, wscale=1.0, keep_v=False, **epi):
    O, I = w.shape[:2]
    x = big[:, :I]
    if epi.get("in_scale") is not None:
        x = x * epi["in_scale"][:, :I, None, None]
    z = F.conv2d(x, w * wscale, stride=spec.stride, padding=spec.pad)
    y = _epilogue(z, _cpad(O, big.dtype), **epi)
    return (y, None) if keep_v else y


def conv_bwd_data(small, w, spec, big_hw, wscale=1.0, **epi):
    O, I = w.shape[:2]
    x = small[:, :O]
    if epi.get("in_scale") is not None:
        x = x * epi["in_scale"][:, :O, None, None]
    Hs, Ws = small.shape[2:]
    op = (big_hw[0] - ((Hs - 1) * spec.stride + spec.KH - 2 * spec.pad), big_hw[1] - ((Ws - 1) * spec.stride + spec.KW - 2 * spec.pad))
    z = F.conv_transpose2d(x, w * wscale, stride=spec.stride, padding=spec.pad, output_padding=op)
    return _epilogue(z, _cpad(I, small.dtype), **epi)


def conv_wgrad(small, big, spec, O, I, wscale=1.0, small_scale=None, big_scale=None, big_v=None):
    gy, x = small[:, :O], big[:, :I]
    if small_scale is not None:
        gy = gy * small_scale[:, :O, None, None]
    if big_scale is not None:
        x = x * big_scale[:, :I, None, None]
    with torch.enable_grad():
        w = torch.zeros(O, I, spec.KH, spec.KW, requires_grad=True)
        y = F.conv2d(x.detach(), w, stride=spec.stride, padding=spec.pad)
        (gw,) = torch.autograd.grad(y, w, gy.detach())
    return gw * wscale


def upfirdn2d(x, k, up, down, pad0, out_hw, flip=True, bias=None, residual=None, act=False, slope=0.2, gain=2 ** 0.5, fuse=None):
    B, C, H, W = x.shape
    KH, KW = k.shape
    Ho, Wo = out_hw
    z = x.reshape(B * C, 1, H, 1, W, 1)
    z = F.pad(z, (0, up - 1, 0, 0, 0, up - 1)).reshape(B * C, 1, H * up, W * up)
    Hp, Wp = (Ho - 1) * down + KH, (Wo - 1) * down + KW  # extent of the padded canvas the taps touch
    z = F.pad(z, (pad0, Wp - (W * up + pad0), pad0, Hp - (H * up + pad0)))
    kf = torch.flip(k, [0, 1]) if flip else k
    z = F.conv2d(z, kf.reshape(1, 1, KH, KW), stride=down)
    z = z.reshape(B, C, Ho, Wo)
    return _epilogue(z, C, bias=bias, residual=residual, act=act, slope=slope, gain=gain, fuse=fuse)


def bias_act(x, bias=None, residual=None, slope=0.2, gain=2 ** 0.5):
    return _epilogue(x, x.shape[1], bias=bias, residual=residual, act=True, slope=slope, gain=gain)


def bias_act_bwd(gy, y, want_gbias, slope=0.2, gain=2 ** 0.5):
    gx = (gy * gain * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))).contiguous(memory_format=CL)
    return gx, (gx.sum(dim=(0, 2, 3)) if want_gbias else None)


def colsum(x):
    return x.sum(dim=(0, 2, 3))


def mul_reduce(a, b, scale=None, want_scaled=False):
    out = (a * b).sum(dim=(2, 3))
    scaled = (scale[:, :, None, None] * a).contiguous(memory_format=CL) if want_scaled else None
    return out, scaled


def act_inv_mul_reduce(g, y, residual, bias, slope, gain):
    z = torch.where(y > 0, y / gain, y / (gain * slope))
    if residual is not None:
        z = z - residual
    if bias is not None:
        z = z - bias[None, :, None, None]
    return (g * z).sum(dim=(2, 3))


def bilinear_down(x, S, backward_to=None):
    if backward_to is None:
        return F.interpolate(x, size=(S, S), mode="bilinear", align_corners=False).contiguous(memory_format=CL)
    R = backward_to
    with torch.enable_grad():
        full = torch.zeros(x.shape[0], x.shape[1], R, R, requires_grad=True)
        y = F.interpolate(full, size=(S, S), mode="bilinear", align_corners=False)
        (g,) = torch.autograd.grad(y, full, x.detach())
    return g.contiguous(memory_format=CL)


def resize(x, out_hw, mode, backward_to=None):
    if backward_to is None:
        return F.interpolate(x, size=tuple(out_hw), mode=mode, align_corners=False)
    with torch.enable_grad():
        full = torch.zeros(x.shape[0], x.shape[1], *backward_to, requires_grad=True)
        y = F.interpolate(full, size=tuple(x.shape[2:]), mode=mode, align_corners=False)
        (g,) = torch.autograd.grad(y, full, x.detach())
    return g


def _mbstd_stat(x, G):
    B, C, H, W = x.shape
    s = x.reshape(G, B // G, C, H, W)
    return torch.sqrt(s.var(0, unbiased=False) + 1e-8).mean(dim=(1, 2, 3))  # [M]


def mbstd_fwd(x, G, Cy):
    B, C, H, W = x.shape
    stat = _mbstd_stat(x, G)
    y = torch.zeros(B, Cy, H, W)
    y[:, :C] = x
    y[:, C] = stat.repeat(G)[:, None, None]
    return y.contiguous(memory_format=CL), stat


def mbstd_bwd(x, gy, G):
    C = x.shape[1]
    with torch.enable_grad():
        xd = x.detach().requires_grad_(True)
        stat = _mbstd_stat(xd, G)
        gstat = gy[:, C].reshape(G, stat.shape[0], -1).sum(dim=(0, 2))
        (gx,) = torch.autograd.grad(stat, xd, gstat.detach())
    return (gx + gy[:, :C]).contiguous(memory_format=CL)


def sqnorm_per_sample(g):
    return g.reshape(g.shape[0], -1).pow(2).sum(dim=1)


def linear_nt(a, b, bias=None, scale=1.0, act=False, slope=0.2, gain=1.0, n_pad=None):
    N, K = b.shape
    n_pad = N if n_pad is None else n_pad
    y = F.pad(scale * (a[:, :K] @ b.t()), (0, n_pad - N))
    valid = torch.arange(n_pad) < N
    if bias is not None:
        y = y + bias[None, :] * valid
    if act:
        y = gain * F.leaky_relu(y, slope)
    return y * valid  # padding columns are written as zero


def linear_nn(a, b, scale=1.0, n_valid=None, k_pad=None):
    N, K = b.shape
    k_pad = K if k_pad is None else k_pad
    return F.pad(scale * (a[:, :N] @ b), (0, k_pad - K))


def linear_tn(a, b, scale=1.0, n_valid=None, k_valid=None):
    N = a.shape[1] if n_valid is None else n_valid
    K = b.shape[1] if k_valid is None else k_valid
    return scale * (a[:, :N].t() @ b[:, :K])


def weight_sq_sum(w):
    return w.pow(2).sum(dim=(2, 3))


def style_demod(s, wsq, scale2, eps, cout_pad):
    cout, cin = wsq.shape
    d = torch.rsqrt(scale2 * (s[:, :cin].pow(2) @ wsq.t()) + eps)
    return F.pad(d, (0, cout_pad - cout), value=1.0)


def _g_acc(gd, d, cout, scale2):
    return gd[:, :cout] * (-0.5 * scale2) * d[:, :cout].pow(3)


def style_demod_bwd_s(gd, d, wsq, s, gs_in, scale2):
    cout, cin = wsq.shape
    g = F.pad(2.0 * s[:, :cin] * (_g_acc(gd, d, cout, scale2) @ wsq), (0, s.shape[1] - cin))
    return g if gs_in is None else g + gs_in


def style_demod_bwd_w(gd, d, s, cout, cin, scale2):
    return _g_acc(gd, d, cout, scale2).t() @ s[:, :cin].pow(2)


def demod_wgrad(w, g_wsq):
    return 2.0 * w * g_wsq[:, :, None, None]


def pack_nhwc(src0, off0, src1, off1, cp, dtype):
    B, _, H, W = src0.shape
    out = torch.zeros(B, cp, H, W, dtype=torch.float32)
    out[:, off0:off0 + src0.shape[1]] = src0
    if src1 is not None:
        out[:, off1:off1 + src1.shape[1]] = src1
    return out.to(dtype).contiguous(memory_format=CL)


def unpack_nhwc(g, c_off, C):
    return g[:, c_off:c_off + C].float().contiguous(memory_format=CL)


def linear_bank_fwd(x, weights, biases, scale):
    K = weights[0].shape[1]
    return [scale * (x[:, :K] @ w.t()) + (0 if b is None else b) for w, b in zip(weights, biases)]


def linear_bank_bwd(x, weights, grads, scale, want_x, want_w, want_b, x_cols=None):
    K = weights[0].shape[1]
    x_cols = K if x_cols is None else x_cols
    gx = F.pad(scale * sum(g @ w for g, w in zip(grads, weights)), (0, x_cols - K)) if want_x else None
    gws = [scale * (g.t() @ x[:, :K]) for g in grads] if want_w else None
    gbs = [g.sum(0) for g in grads] if want_b else None
    return gx, gws, gbs


OPS = dict(linear_bank_fwd=linear_bank_fwd, linear_bank_bwd=linear_bank_bwd, pack_nhwc=pack_nhwc, unpack_nhwc=unpack_nhwc, weight_sq_sum=weight_sq_sum, style_demod=style_demod, style_demod_bwd_s=style_demod_bwd_s, style_demod_bwd_w=style_demod_bwd_w,
           demod_wgrad=demod_wgrad, resize=resize, linear_nt=linear_nt, linear_nn=linear_nn, linear_tn=linear_tn, nhwc=nhwc, conv_fwd=conv_fwd, conv_bwd_data=conv_bwd_data, conv_wgrad=conv_wgrad, upfirdn2d=upfirdn2d,
           bias_act=bias_act, bias_act_bwd=bias_act_bwd, colsum=colsum, mul_reduce=mul_reduce,
           act_inv_mul_reduce=act_inv_mul_reduce, bilinear_down=bilinear_down, mbstd_fwd=mbstd_fwd, mbstd_bwd=mbstd_bwd,
           sqnorm_per_sample=sqnorm_per_sample)


def install(monkeypatch=None):
    """Replace the HIP launches of gif_amd.ops by their ATen restatements (undone by monkeypatch, or call uninstall)."""
    from gif_amd import ops
    saved = {k: getattr(ops, k) for k in OPS}
    for k, v in OPS.items():
        if monkeypatch is not None:
            monkeypatch.setattr(ops, k, v)
        else:
            setattr(ops, k, v)
    return saved
