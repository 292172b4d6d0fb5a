"""torch.autograd bindings of the gfx950 kernels.

Two families:
  * Conv2dFn / WgradFn / BiasActFn / Upfirdn2dFn / MbstdFn — every backward is itself expressed with these
    Functions, so they are differentiable to any order.  The discriminator uses them because R1
    (grad_penalty_loss, loss_functions/losses.py:87-99, create_graph=True) back-propagates through D's backward.
    ConvBiasActFn / BlurBiasActFn fuse bias (+ condition noise) + leaky ReLU into the producing kernel's epilogue
    and keep that property (their backward is composed of the Functions above).
  * ModConvFn / ModConvActFn — the generator's modulated convolution (ModulatedConv2d.forward,
    stylegan2_common_layers.py:307-349) as ONE fused op: y = [act](d[b,co] * conv(s[b,ci] * x, W) [+ noise + bias]).
    The first-order backward is a handful of raw fused launches.  When the backward itself is recorded
    (create_graph=True: the StyleGAN2-form path-length regulariser, DIRECT_GRAD_REG of train.py:209-215) the gradients are
    instead produced by differentiating a composition of the any-order Functions above (_modconv_composite), so the
    generator is differentiable to any order as well — slower, but never silently wrong.
  * BilinearDownFn — the condition pyramid; _TextureMapFn lives in gif_amd/texture_space.py.
All tensors are logical NCHW with NHWC memory (ops.nhwc).
"""
import torch
from torch.autograd import Function

from . import ops
from .ops import ConvSpec


# --------------------------------------------------------------------------------------------------------
# Activation ports: FusedLeakyReLU's backward inside the kernel that PRODUCES the gradient
# --------------------------------------------------------------------------------------------------------
# A layer whose epilogue applies bias + leaky ReLU (ConvBiasActFn, ModConvActFn, BlurBiasActFn) hands out, next to its
# output y, an ALIAS of y as a second autograd output — the layer's "port" — and tags y with it (y._gif_port).  A consumer
# that knows the protocol (the convolutions and the blur below) takes BOTH y and the alias as autograd inputs and, in a
# first-order backward, returns nothing for y and, for the alias, the gradient ALREADY multiplied by gain * (y > 0 ? 1 : slope):
# the mask is applied in the epilogue of the data-gradient / FIR kernel that produces that gradient (ops.GradFuse; the
# consumer has y saved as its own input), the bias-gradient column sums come out of the same epilogue and travel through
# `parts`.  The producing layer then finds its pre-activation gradient ready-made on the port and skips the stand-alone pass
# (reference: FusedLeakyReLUFunctionBackward, stylegan2_common_layers.py:22-39, as its own kernel: read g, read y, write g' —
# three full-resolution tensors per layer).  Gradients from consumers that do not know the protocol arrive on y as before and
# take the stand-alone pass; both add up.  Under create_graph (R1, the path-length / direct-gradient regularisers) consumers
# return the plain, differentiable gradient on y and leave the port alone: a port only ever carries first-order gradients.
class ActPort:
    __slots__ = ("alias", "slope", "gain", "want_bias", "parts")

    def __init__(self, slope, gain, want_bias):
        self.alias, self.slope, self.gain, self.want_bias = None, float(slope), float(gain), bool(want_bias)
        self.parts = []  # [C] bias-gradient contributions appended by the consumers' backward passes

    def cfg(self):
        """What a consumer's backward needs (no reference to the alias: the autograd node must not own a cycle)."""
        return (self.slope, self.gain, self.want_bias, self.parts)


def _take_port(x, identity=False):
    """(port alias or None, port cfg or None) for a consumer's input x.  identity selects the kind of port the consumer can
    serve: False = a real activation (mask in the gradient kernel's epilogue), True = bias-only layers (residual inputs)."""
    port = getattr(x, "_gif_port", None)
    if (port is None or not ops.FUSE_GRAD or not torch.is_grad_enabled() or port.alias is None or not port.alias.requires_grad
            or (port.slope == 1.0 and port.gain == 1.0) != identity):
        return None, None
    return port.alias, port.cfg()


# --------------------------------------------------------------------------------------------------------
# inner gradients of the regularisers: no parameter gradients
# --------------------------------------------------------------------------------------------------------
# R1 (losses.grad_penalty_loss), the path-length and the direct-gradient regulariser take torch.autograd.grad(..., inputs=[image /
# style / condition], create_graph=True).  ctx.needs_input_grad of a custom Function is fixed at forward time, so every node
# would also compute the gradient of its weight and bias in that pass — D's complete weight gradient, thrown away by the
# engine.  Inside inputs_only_backward() the Functions below skip gradients of inputs that ARE parameters (or views of one).
_inputs_only = False


class inputs_only_backward:
    def __enter__(self):
        global _inputs_only
        self._prev, _inputs_only = _inputs_only, True

    def __exit__(self, *exc):
        global _inputs_only
        _inputs_only = self._prev


def _param_mask(*tensors):
    P = torch.nn.Parameter
    return tuple(isinstance(t, torch.Tensor) and (isinstance(t, P) or isinstance(getattr(t, "_base", None), P)) for t in tensors)


def _need(ctx, i):
    """Input i needs a gradient in THIS backward pass."""
    return ctx.needs_input_grad[i] and not (_inputs_only and ctx.pmask[i])


def _port_on(ctx):
    """This backward delivers to the input's port (first-order pass) instead of to the input itself."""
    return ctx.in_port is not None and not torch.is_grad_enabled()


def _tag(y, alias, port):
    if port is not None:
        port.alias = alias
        y._gif_port = port
    return y


def _new_port(x_requires_grad, slope, gain, bias):
    """Port of an activation layer's output, or None when nothing will be back-propagated through it."""
    if not ops.FUSE_GRAD or not torch.is_grad_enabled():
        return None
    want_bias = bias is not None and bias.requires_grad
    if slope == 1.0 and gain == 1.0 and not want_bias:
        return None  # identity activation without a bias gradient: nothing a consumer could take over
    return ActPort(slope, gain, want_bias)


def _deliver_residual(r_port, gpre, gb):
    """A layer's `residual` input is the output of an identity-activation layer (the last condition-noise conv: conv + bias):
    its gradient IS this layer's pre-activation gradient gpre, and its bias gradient is the same column sum gb this layer takes
    for its own bias — both are handed over (first-order passes) instead of the producer launching a column-sum pass over gpre."""
    _, _, want_b, parts = r_port
    if want_b:
        parts.append(gb if gb is not None else ops.colsum(gpre))
    return gpre


def _mask_into_port(cfg, gx, x):
    """Stand-alone form of a consumer's duty (differentiable to any order): mask gx with the activation of x, file the bias part."""
    slope, gain, want_b, parts = cfg
    gxm, gb = BiasActBwdFn.apply(gx, x, want_b, slope, gain)
    if want_b:
        parts.append(gb)
    return gxm


def _producer_gpre(gy, g_port, y, want_b, slope, gain, parts):
    """Pre-activation gradient (and bias gradient) of an activation layer from what arrived on y (unmasked) and on its port
    (masked by the consumers)."""
    gpre = gb = None
    if gy is not None:
        gpre, gb = BiasActBwdFn.apply(gy, y, want_b, slope, gain)
        if not want_b:
            gb = None
    if g_port is not None:
        gpre = g_port if gpre is None else gpre + g_port
        if want_b and parts:
            for part in parts:
                gb = part if gb is None else gb + part
    if parts is not None:
        parts.clear()
    return gpre, gb


# --------------------------------------------------------------------------------------------------------
# plain convolution family (twice differentiable)
# --------------------------------------------------------------------------------------------------------
class Conv2dFn(Function):
    """transposed=False: y = conv2d(x, w*wscale); True: y = conv_transpose2d(x, w*wscale) to size out_hw; optional
    `residual` (same shape as y) is added in the kernel's epilogue.  w is the canonical FORWARD-conv weight [O,I,KH,KW]
    in both cases."""

    @staticmethod
    def forward(ctx, x, w, spec, transposed, out_hw, wscale, residual=None):
        ctx.pmask = _param_mask(x, w)
        x = ops.nhwc(x)
        ctx.spec, ctx.transposed, ctx.wscale = spec, transposed, wscale
        ctx.save_for_backward(x, w)
        ctx.v = None
        epi = {} if residual is None else {"residual": ops.nhwc(residual)}
        if not transposed:
            # keep the Winograd-transformed input for the weight gradient (same x): saves one HBM-bound transform pass
            y, v = ops.conv_fwd(x, w, spec, wscale, keep_v=True, **epi)
            ctx.v = v if ctx.needs_input_grad[1] else None
            return y
        return ops.conv_bwd_data(x, w, spec, tuple(out_hw), wscale, **epi)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        spec, tr, ws = ctx.spec, ctx.transposed, ctx.wscale
        gx = gw = gr = None
        if _need(ctx, 0):
            gx = Conv2dFn.apply(gy, w, spec, not tr, tuple(x.shape[2:]), ws, None)
            assert gx.shape == x.shape, (gx.shape, x.shape)
        if _need(ctx, 1):
            O, I = w.shape[:2]
            gw = WgradFn.apply(gy, x, spec, O, I, ws, ctx.v) if not tr else WgradFn.apply(x, gy, spec, O, I, ws)
        if ctx.needs_input_grad[6]:
            gr = gy
        return gx, gw, None, None, None, None, gr


class WgradFn(Function):
    """dW[O,I,KH,KW] = wscale * sum_{b,pix} small (x) big   (small = conv-output side, big = conv-input side)."""

    @staticmethod
    def forward(ctx, small, big, spec, O, I, wscale, big_v=None):
        small, big = ops.nhwc(small), ops.nhwc(big)
        ctx.spec, ctx.wscale = spec, wscale
        ctx.save_for_backward(small, big)
        return ops.conv_wgrad(small, big, spec, O, I, wscale, big_v=big_v)

    @staticmethod
    def backward(ctx, ggw):
        small, big = ctx.saved_tensors
        gs = gb = None
        if ctx.needs_input_grad[0]:
            gs = Conv2dFn.apply(big, ggw, ctx.spec, False, None, ctx.wscale, None)
        if ctx.needs_input_grad[1]:
            gb = Conv2dFn.apply(small, ggw, ctx.spec, True, tuple(big.shape[2:]), ctx.wscale, None)
        return gs, gb, None, None, None, None, None


class ConvBiasActFn(Function):
    """y = gain*lrelu(conv2d(x, w*wscale) + bias): bias and activation run in the conv kernel's epilogue (ConvLayer =
    EqualConv2d + FusedLeakyReLU, stylegan2_common_layers.py:752-799).  The backward is composed of BiasActBwdFn,
    Conv2dFn and WgradFn, hence differentiable to any order (R1); its first-order form fuses the activation backward of the
    layer that produced x into the data-gradient kernel (in_port, see ActPort).

    Outputs (y, port alias of y, x'): x' (passthrough=True, else None) is an alias of x for a SECOND consumer of x (the
    ResBlock's skip branch): that consumer's gradient then arrives here, in this node's backward, and is added by the
    data-gradient kernel's epilogue (`residual`) instead of by a separate gradient-accumulation pass over two
    full-resolution tensors."""

    @staticmethod
    def forward(ctx, x, w, bias, spec, wscale, slope, gain, passthrough=False, in_port=None, parts=None, x_port=None):
        ctx.set_materialize_grads(False)  # an unused output (y, its port or the alias) arrives as None, not as a zero tensor
        ctx.pmask = _param_mask(x, w, bias)
        x = ops.nhwc(x)
        y, v = ops.conv_fwd(x, w, spec, wscale, keep_v=True, bias=bias, act=True, slope=slope, gain=gain)
        ctx.v = v if ctx.needs_input_grad[1] else None  # Winograd-transformed x, reused by the weight gradient
        ctx.cfg = (spec, wscale, slope, gain, bias is not None)
        ctx.in_port, ctx.parts = in_port, parts
        ctx.save_for_backward(x, w, y)
        # the port alias is y.detach(), NOT a view of y: a view keeps its base alive, and y's Python object (kept alive by the
        # C++ tensor once it carries the _gif_port attribute) references the alias through the port — with a view that is a
        # reference cycle through C++ that no collector sees, and every iteration's graph would leak (tests/test_cpu_wiring.py)
        return y, y.detach(), (x.view_as(x) if passthrough else None)

    @staticmethod
    def backward(ctx, gy, g_port=None, g_alias=None):
        x, w, y = ctx.saved_tensors
        spec, ws, slope, gain, has_bias = ctx.cfg
        want_b = has_bias and _need(ctx, 2)
        gx = gw = None
        port = _port_on(ctx)
        gpre, gb = _producer_gpre(gy, g_port, y, want_b, slope, gain, ctx.parts)
        if gpre is not None:
            if ctx.needs_input_grad[0]:
                gx = _conv_dgrad(gpre, w, spec, tuple(x.shape[2:]), ws, g_alias, ctx.in_port if port else None, x)
            if _need(ctx, 1):
                gw = WgradFn.apply(gpre, x, spec, w.shape[0], w.shape[1], ws, ctx.v)
        elif ctx.needs_input_grad[0] and g_alias is not None:
            gx = _mask_into_port(ctx.in_port, g_alias, x) if port else g_alias
        return (None if port else gx), gw, (gb if want_b else None), None, None, None, None, None, None, None, (gx if port else None)


def _conv_dgrad(gpre, w, spec, big_hw, ws, residual, in_port, x):
    """Data gradient of conv2d(x, w) (+ residual); with in_port (first-order passes only) the result is masked by the activation
    that produced x, and the bias-gradient column sums are taken, in the kernel's epilogue."""
    if in_port is None:
        return Conv2dFn.apply(gpre, w, spec, True, big_hw, ws, residual)
    slope, gain, want_b, parts = in_port
    fuse = ops.GradFuse(mask_src=x, mask_slope=slope, mask_gain=gain, want_colsum=want_b)
    gx = ops.conv_bwd_data(gpre, w, spec, big_hw, ws, residual=None if residual is None else ops.nhwc(residual), fuse=fuse)
    if want_b:
        parts.append(fuse.colsum)
    return gx


def conv2d_bias_act(x, w, bias, stride=1, pad=0, wscale=1.0, slope=0.2, gain=2 ** 0.5, passthrough=False):
    """passthrough=True: returns (y, alias of x) — see ConvBiasActFn."""
    x_port, in_port = _take_port(x)
    port = _new_port(True, slope, gain, bias)
    y, alias, x_alias = ConvBiasActFn.apply(x, w, bias, ConvSpec(w.shape[2], w.shape[3], stride, pad), wscale, slope, gain,
                                            bool(passthrough), in_port, None if port is None else port.parts, x_port)
    _tag(y, alias, port)
    return (y, x_alias) if passthrough else y


def conv2d(x, w, stride=1, pad=0, wscale=1.0, residual=None):
    """x [B,C,H,W] with C == pad4(w.shape[1]); returns [B, pad4(O), Ho, Wo] (+ residual, added in the kernel epilogue)."""
    return Conv2dFn.apply(x, w, ConvSpec(w.shape[2], w.shape[3], stride, pad), False, None, wscale, residual)


def conv_transpose2d(x, w, stride, pad, out_hw, wscale=1.0):
    """Adjoint of conv2d(., w): x [B, pad4(O), Hs, Ws] -> [B, pad4(I), *out_hw]."""
    return Conv2dFn.apply(x, w, ConvSpec(w.shape[2], w.shape[3], stride, pad), True, tuple(out_hw), wscale, None)


# --------------------------------------------------------------------------------------------------------
# linear layers (EqualLinear, stylegan2_common_layers.py:193-235) on the skinny-GEMM kernels (csrc/linear.hip).
# Three products, closed under differentiation like Conv2dFn / WgradFn (R1 differentiates the discriminator head twice, the
# path-length regulariser the mapping network):  nt: a @ b^T   nn: a @ b   tn: a^T @ b.   Column padding is explicit: an
# operand may carry zero columns beyond the logical width, outputs are produced with the requested padded width.
# --------------------------------------------------------------------------------------------------------
class LinearNtFn(Function):
    """y [M, n_pad] = scale * a[:, :K] @ b[N, K]^T  (columns N..n_pad zero)."""

    @staticmethod
    def forward(ctx, a, b, scale, n_pad):
        ctx.scale = scale
        ctx.pmask = _param_mask(a, b)
        ctx.save_for_backward(a, b)
        return ops.linear_nt(a, b, None, scale, n_pad=n_pad)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        N, K = b.shape
        ga = LinearNnFn.apply(g, b, ctx.scale, a.shape[1]) if _need(ctx, 0) else None
        gb = LinearTnFn.apply(g, a, ctx.scale, N, K) if _need(ctx, 1) else None
        return ga, gb, None, None


class LinearNnFn(Function):
    """y [M, k_pad] = scale * a[:, :N] @ b[N, K]  (columns K..k_pad zero)."""

    @staticmethod
    def forward(ctx, a, b, scale, k_pad):
        ctx.scale = scale
        ctx.pmask = _param_mask(a, b)
        ctx.save_for_backward(a, b)
        return ops.linear_nn(a, b, scale, k_pad=k_pad)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        N, K = b.shape
        ga = LinearNtFn.apply(g, b, ctx.scale, a.shape[1]) if _need(ctx, 0) else None
        gb = LinearTnFn.apply(a, g, ctx.scale, N, K) if _need(ctx, 1) else None
        return ga, gb, None, None


class LinearTnFn(Function):
    """y [N, K] = scale * a[:, :N]^T @ b[:, :K]."""

    @staticmethod
    def forward(ctx, a, b, scale, N, K):
        ctx.scale = scale
        ctx.pmask = _param_mask(a, b)
        ctx.save_for_backward(a, b)
        return ops.linear_tn(a, b, scale, n_valid=N, k_valid=K)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = LinearNtFn.apply(b, g, ctx.scale, a.shape[1]) if _need(ctx, 0) else None
        gb = LinearNnFn.apply(a, g, ctx.scale, b.shape[1]) if _need(ctx, 1) else None
        return ga, gb, None, None, None


class LinearBiasActFn(Function):
    """y [M, n_pad] = gain * lrelu(scale * x @ w^T + bias)  (act) or scale * x @ w^T + bias: one launch, bias and activation in
    the GEMM epilogue; backward composed of BiasActBwdFn + LinearNnFn + LinearTnFn (any order)."""

    @staticmethod
    def forward(ctx, x, w, bias, scale, act, slope, gain, n_pad):
        y = ops.linear_nt(x, w, bias, scale, act=act, slope=slope, gain=gain, n_pad=n_pad)
        ctx.pmask = _param_mask(x, w, bias)
        ctx.cfg = (scale, act, slope, gain, bias is not None)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        scale, act, slope, gain, has_bias = ctx.cfg
        N, K = w.shape
        want_b = has_bias and _need(ctx, 2)
        M, n_pad = y.shape
        if act:
            gpre, gb = BiasActBwdFn.apply(gy.reshape(M, n_pad, 1, 1), y.reshape(M, n_pad, 1, 1), want_b, slope, gain)
        else:
            gpre, gb = BiasActBwdFn.apply(gy.reshape(M, n_pad, 1, 1), y.reshape(M, n_pad, 1, 1), want_b, 1.0, 1.0)
        gpre = gpre.reshape(M, n_pad)
        gx = LinearNnFn.apply(gpre, w, scale, x.shape[1]) if _need(ctx, 0) else None
        gw = LinearTnFn.apply(gpre, x, scale, N, K) if _need(ctx, 1) else None
        return gx, gw, (gb if want_b else None), None, None, None, None, None


def linear_bias_act(x, w, bias=None, scale=1.0, act=False, slope=0.2, gain=1.0, n_pad=None):
    """x [M, K'] (K' >= K = w.shape[1], K % 4 == 0), w [N, K], bias [n_pad] or None -> [M, n_pad]."""
    n_pad = w.shape[0] if n_pad is None else n_pad
    if bias is None and not act:
        return LinearNtFn.apply(x, w, scale, n_pad)
    return LinearBiasActFn.apply(x, w, bias, scale, bool(act), slope, gain, n_pad)


class LinearBankFn(Function):
    """(scale * x @ w_l^T + b_l for every layer l) in ONE launch; backward = one launch for all weight and bias gradients plus
    one for the input gradient (ops.linear_bank_*).  While a backward is being recorded (the path-length regulariser
    differentiates the modulation twice) the gradients are composed of the any-order pieces LinearNnFn / LinearTnFn instead."""

    @staticmethod
    def forward(ctx, x, scale, n, *wb):
        ws, bs = wb[:n], wb[n:]
        ctx.scale, ctx.n = scale, n
        ctx.has_bias = [b is not None for b in bs]
        ctx.save_for_backward(x, *ws)
        return tuple(ops.linear_bank_fwd(x, ws, bs, scale))

    @staticmethod
    def backward(ctx, *gs):
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        n, scale = ctx.n, ctx.scale
        live = [i for i in range(n) if gs[i] is not None]
        need_x = ctx.needs_input_grad[0]
        need_w = [ctx.needs_input_grad[3 + i] for i in range(n)]
        need_b = [ctx.has_bias[i] and ctx.needs_input_grad[3 + n + i] for i in range(n)]
        gx, gw, gb = None, [None] * n, [None] * n
        if not live:
            return (None, None, None) + tuple(gw) + tuple(gb)
        if torch.is_grad_enabled():  # create_graph=True: stay differentiable
            for i in live:
                if need_x:
                    t = LinearNnFn.apply(gs[i], ws[i], scale, x.shape[1])
                    gx = t if gx is None else gx + t
                if need_w[i]:
                    gw[i] = LinearTnFn.apply(gs[i], x, scale, ws[i].shape[0], ws[i].shape[1])
                if need_b[i]:
                    gb[i] = gs[i].sum(0)
        else:
            want_w, want_b = any(need_w[i] for i in live), any(need_b[i] for i in live)
            gx, gws, gbs = ops.linear_bank_bwd(x, [ws[i] for i in live], [gs[i] for i in live], scale, need_x, want_w, want_b,
                                               x_cols=x.shape[1])
            for j, i in enumerate(live):
                if need_w[i]:
                    gw[i] = gws[j]
                if need_b[i]:
                    gb[i] = gbs[j]
        return (gx, None, None) + tuple(gw) + tuple(gb)


def linear_bank(x, weights, biases, scale):
    """[scale * x @ w^T + b for (w, b) in zip(weights, biases)] — x [M, K'], w [n, K] (n % 8 == 0), b [n] or None."""
    return LinearBankFn.apply(x, float(scale), len(weights), *weights, *biases)


# --------------------------------------------------------------------------------------------------------
# style path of ModulatedConv2d (stylegan2_common_layers.py:311-320): demodulation d = rsqrt(scale^2 * s^2 @ wsq^T + eps)
# --------------------------------------------------------------------------------------------------------
def _demod_reference(s, w, scale2, eps, cout_pad):
    """The same function out of any-order pieces (LinearNtFn + torch pointwise ops): what a recorded backward differentiates."""
    wsq = w.pow(2).sum(dim=(2, 3))
    cin = wsq.shape[1]
    cout = wsq.shape[0]
    acc = LinearNtFn.apply(s[:, :cin].pow(2), wsq, scale2, ops.pad4(cout))[:, :cout]  # (4-float row granularity of the GEMMs)
    d = torch.rsqrt(acc + eps)
    return d if cout_pad == d.shape[1] else torch.nn.functional.pad(d, (0, cout_pad - d.shape[1]), value=1.0)


class DemodFn(Function):
    """d [B, cout_pad] from the modulation s [B, cin_pad] and the conv weight w [Cout, Cin, k, k]: ONE skinny GEMM with the
    square of s in its operand load and rsqrt in its epilogue (+ wsq = sum_taps w^2, cached per weight version); backward = two
    GEMMs with the chain-rule factors in their operand loads and one pass over the weight.  The reference (and round 2) spent ~10
    tiny ATen launches per layer and direction on this.  A recorded backward (create_graph) differentiates _demod_reference."""

    @staticmethod
    def forward(ctx, s, w, scale2, eps, cout_pad):
        wsq = ops.weight_sq_sum(w)
        d = ops.style_demod(s, wsq, scale2, eps, cout_pad)
        ctx.pmask = _param_mask(s, w)
        ctx.cfg = (scale2, eps, cout_pad)
        ctx.save_for_backward(s, w, d)
        return d

    @staticmethod
    def backward(ctx, gd):
        s, w, d = ctx.saved_tensors
        scale2, eps, cout_pad = ctx.cfg
        if torch.is_grad_enabled():
            gs, gw = _recorded_backward((s, w), (_need(ctx, 0), _need(ctx, 1)), gd,
                                        lambda s_, w_: _demod_reference(s_, w_, scale2, eps, cout_pad))
            return gs, gw, None, None, None
        gs = gw = None
        cout, cin = w.shape[:2]
        gd = gd.contiguous()
        if ctx.needs_input_grad[0]:
            gs = ops.style_demod_bwd_s(gd, d, ops.weight_sq_sum(w), s, None, scale2)
        if _need(ctx, 1):
            gw = ops.demod_wgrad(w, ops.style_demod_bwd_w(gd, d, s, cout, cin, scale2))
        return gs, gw, None, None, None


def demodulation(s, w, scale, eps, cout_pad):
    return DemodFn.apply(s, w, float(scale) ** 2, float(eps), int(cout_pad))


# --------------------------------------------------------------------------------------------------------
# fused bias + leaky ReLU (FusedLeakyReLU, stylegan2_common_layers.py:22-39)
# --------------------------------------------------------------------------------------------------------
class BiasActFn(Function):
    @staticmethod
    def forward(ctx, x, bias, residual, slope, gain):
        y = ops.bias_act(x, bias, residual, slope, gain)
        ctx.pmask = _param_mask(x, bias, residual)
        ctx.slope, ctx.gain = slope, gain
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        want_b = ctx.has_bias and _need(ctx, 1)
        gx, gb = BiasActBwdFn.apply(gy, y, want_b, ctx.slope, ctx.gain)
        return (gx if ctx.needs_input_grad[0] else None, gb if want_b else None,
                gx if (ctx.has_res and ctx.needs_input_grad[2]) else None, None, None)


class BiasActBwdFn(Function):
    """gx = gy * gain * (y>0 ? 1 : slope); gbias = sum_pix gx.  Linear in gy; the mask is locally constant in y."""

    @staticmethod
    def forward(ctx, gy, y, want_gbias, slope, gain):
        if slope == 1.0 and gain == 1.0:
            # identity activation (bias only, e.g. the last condition-noise conv): gx IS gy — no elementwise pass, the bias
            # gradient is one column-sum read of gy
            gx = ops.nhwc(gy).view_as(gy)
            gb = ops.colsum(gy) if want_gbias else None
        else:
            gx, gb = ops.bias_act_bwd(gy, y, want_gbias, slope, gain)
        ctx.slope, ctx.gain = slope, gain
        ctx.save_for_backward(y)
        if gb is None:
            gb = gx.new_empty(())  # 0-dim placeholder (never read): no fill kernel
            ctx.mark_non_differentiable(gb)
        return gx, gb

    @staticmethod
    def backward(ctx, ggx, ggb):
        (y,) = ctx.saved_tensors
        t = ggx
        if ggb is not None and ggb.dim() == 1:
            ggb = ggb.to(y.dtype)
            t = ggb.view(1, -1, 1, 1).expand_as(y) if t is None else t + ggb.view(1, -1, 1, 1)
        if t is None:
            return None, None, None, None, None
        ggy, _ = BiasActBwdFn.apply(t, y, False, ctx.slope, ctx.gain)
        return ggy, None, None, None, None


def bias_act(x, bias=None, residual=None, slope=0.2, gain=2 ** 0.5):
    """gain * leaky_relu(x + residual + bias[c], slope); bias is a flat [C] tensor matching x's channel count."""
    return BiasActFn.apply(x, bias, residual, slope, gain)


# --------------------------------------------------------------------------------------------------------
# upfirdn2d (stylegan2_common_layers.py:42-72)
# --------------------------------------------------------------------------------------------------------
class Upfirdn2dFn(Function):
    """in_port: x is the output of a fused leaky ReLU whose backward this op's backward applies to the gradient it produces
    (the blur between ConvLayer conv1 and the stride-2 conv2 of a ResBlock) — see ActPort."""

    @staticmethod
    def forward(ctx, x, k, up, down, pad0, out_hw, flip, in_port=None, x_port=None):
        x = ops.nhwc(x)
        ctx.cfg = (up, down, pad0, flip, tuple(x.shape[2:]))
        ctx.in_port = in_port
        ctx.save_for_backward(k, *((x,) if in_port is not None else ()))
        return ops.upfirdn2d(x, k, up, down, pad0, tuple(out_hw), flip)

    @staticmethod
    def backward(ctx, gy):
        k = ctx.saved_tensors[0]
        up, down, pad0, flip, in_hw = ctx.cfg
        # adjoint of a FIR resampler is a FIR resampler: swap up/down, reverse the taps, pad0' = K-1-pad0
        if not _port_on(ctx):
            return Upfirdn2dFn.apply(gy, k, down, up, k.shape[0] - 1 - pad0, in_hw, not flip), None, None, None, None, None, None, None, None
        x = ctx.saved_tensors[1]
        slope, gain, want_b, parts = ctx.in_port
        if not ops.fir_fusable(x.shape[1], down, up, k.shape, x.dtype):
            gx = _mask_into_port(ctx.in_port, Upfirdn2dFn.apply(gy, k, down, up, k.shape[0] - 1 - pad0, in_hw, not flip), x)
        else:
            fuse = ops.GradFuse(mask_src=x, mask_slope=slope, mask_gain=gain, want_colsum=want_b)
            gx = ops.upfirdn2d(ops.nhwc(gy), k, down, up, k.shape[0] - 1 - pad0, in_hw, not flip, fuse=fuse)
            if want_b:
                parts.append(fuse.colsum)
        return None, None, None, None, None, None, None, None, gx


class BlurBiasActFn(Function):
    """y = gain*lrelu(upfirdn2d(x) + residual + bias): the FIR kernel's epilogue adds the condition-noise, the bias and
    applies the leaky ReLU (StyledConv with upsample: Blur -> NoiseInjection -> FusedLeakyReLU, :322-333, :479-486).
    Outputs (y, port alias of y) — see ActPort."""

    @staticmethod
    def forward(ctx, x, k, pad0, out_hw, residual, bias, slope, gain, parts=None, r_port=None, r_alias=None):
        ctx.set_materialize_grads(False)
        x = ops.nhwc(x)
        y = ops.upfirdn2d(x, k, 1, 1, pad0, tuple(out_hw), True, bias=bias, residual=residual, act=True, slope=slope,
                          gain=gain)
        ctx.cfg = (pad0, tuple(x.shape[2:]), slope, gain, residual is not None, bias is not None)
        ctx.pmask = _param_mask(x, k, None, None, residual, bias)
        ctx.parts, ctx.r_port = parts, r_port
        ctx.save_for_backward(k, y)
        return y, y.detach()  # (not a view of y: see ConvBiasActFn.forward)

    @staticmethod
    def backward(ctx, gy, g_port=None):
        k, y = ctx.saved_tensors
        pad0, in_hw, slope, gain, has_res, has_bias = ctx.cfg
        want_b = has_bias and _need(ctx, 5)
        gpre, gb = _producer_gpre(gy, g_port, y, want_b, slope, gain, ctx.parts)
        gx = gr = gra = None
        if gpre is not None and ctx.needs_input_grad[0]:
            gx = Upfirdn2dFn.apply(gpre, k, 1, 1, k.shape[0] - 1 - pad0, in_hw, False)
        if gpre is not None and has_res and ctx.needs_input_grad[4]:
            if ctx.r_port is not None and not torch.is_grad_enabled():
                gra = _deliver_residual(ctx.r_port, gpre, gb)
            else:
                gr = gpre
        return gx, None, None, None, gr, (gb if want_b else None), None, None, None, None, gra


def blur_bias_act(x, kernel, pad, residual, bias, slope=0.2, gain=2 ** 0.5):
    kh = kernel.shape[0]
    H, W = x.shape[2:]
    out_hw = (H + pad[0] + pad[1] - kh + 1, W + pad[0] + pad[1] - kh + 1)
    port = _new_port(True, slope, gain, bias)
    r_alias, r_port = (None, None) if residual is None else _take_port(residual, identity=True)
    y, alias = BlurBiasActFn.apply(x, kernel, pad[0], out_hw, residual, bias, slope, gain, None if port is None else port.parts,
                                   r_port, r_alias)
    return _tag(y, alias, port)


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Same call signature as the reference function; square kernels, symmetric-axis pads."""
    kh, kw = kernel.shape
    assert kh == kw, "square FIR kernels only"
    H, W = x.shape[2:]
    Ho = (H * up + pad[0] + pad[1] - kh) // down + 1
    Wo = (W * up + pad[0] + pad[1] - kw) // down + 1
    x_port, in_port = _take_port(x)
    return Upfirdn2dFn.apply(x, kernel, up, down, pad[0], (Ho, Wo), True, in_port, x_port)


# --------------------------------------------------------------------------------------------------------
# condition pyramid (stg2_generator.py:309-314)
# --------------------------------------------------------------------------------------------------------
class BilinearDownFn(Function):
    @staticmethod
    def forward(ctx, x, S):
        ctx.cfg = (x.shape[2], S)
        return ops.bilinear_down(x, S)

    @staticmethod
    def backward(ctx, gy):
        R, S = ctx.cfg
        return BilinearDownBwdFn.apply(gy, R, S), None


class BilinearDownBwdFn(Function):  # linear map: its backward is the forward again
    @staticmethod
    def forward(ctx, gy, R, S):
        ctx.cfg = (R, S)
        return ops.bilinear_down(gy, S, backward_to=R)

    @staticmethod
    def backward(ctx, ggx):
        R, S = ctx.cfg
        return BilinearDownFn.apply(ggx, S), None, None


# --------------------------------------------------------------------------------------------------------
# input assembly: cat + channel padding + dtype / layout conversion in one pass (stg2_discriminator.py:48-53)
# --------------------------------------------------------------------------------------------------------
class PackNhwcFn(Function):
    """y [B,cp,H,W] (NHWC, `dtype`) = channels of a (at a_off) and b (at b_off, or None), zero elsewhere.  Linear: its backward
    is UnpackNhwcFn per source, whose backward is PackNhwcFn again — differentiable to any order (R1 differentiates the
    discriminator twice w.r.t. the image, train.py:148)."""

    @staticmethod
    def forward(ctx, a, b, a_off, b_off, cp, dtype):
        ctx.cfg = (a_off, a.shape[1], b_off, None if b is None else b.shape[1])
        return ops.pack_nhwc(a.float(), a_off, None if b is None else b.float(), b_off, cp, dtype)

    @staticmethod
    def backward(ctx, gy):
        a_off, ca, b_off, cb = ctx.cfg
        ga = UnpackNhwcFn.apply(gy, a_off, ca) if ctx.needs_input_grad[0] else None
        gb = UnpackNhwcFn.apply(gy, b_off, cb) if (cb is not None and ctx.needs_input_grad[1]) else None
        return ga, gb, None, None, None, None


class UnpackNhwcFn(Function):
    @staticmethod
    def forward(ctx, g, c_off, C):
        ctx.cfg = (c_off, g.shape[1], g.dtype)
        return ops.unpack_nhwc(g, c_off, C)

    @staticmethod
    def backward(ctx, gg):
        c_off, cp, dtype = ctx.cfg
        return PackNhwcFn.apply(gg, None, c_off, 0, cp, dtype), None, None


def pack_nhwc(a, b=None, cp=None, dtype=torch.float32):
    """cat((a, b), 1) zero-padded to cp channels, as an NHWC tensor of `dtype` (b may be None)."""
    ca = a.shape[1]
    return PackNhwcFn.apply(a, b, 0, ca, cp, dtype)


def bilinear_down(x, S):
    """F.interpolate(x, (S,S), 'bilinear', align_corners=False) for square inputs with R/S == 1 or even."""
    return BilinearDownFn.apply(x, S)


# --------------------------------------------------------------------------------------------------------
# minibatch standard deviation (stg2_discriminator.py:59-65)
# --------------------------------------------------------------------------------------------------------
def _mbstd_torch(x, G):
    """Differentiable torch restatement on the tiny [B,C,4,4] tensor; used ONLY for the second-order term of R1."""
    B, C, H, W = x.shape
    s = x.reshape(G, B // G, C, H, W)
    s = torch.sqrt(s.var(0, unbiased=False) + 1e-8).mean(dim=(1, 2, 3))  # [M]
    return s


class MbstdFn(Function):
    @staticmethod
    def forward(ctx, x, G, Cy):
        x = ops.nhwc(x)
        ctx.G = G
        ctx.save_for_backward(x)
        y, _ = ops.mbstd_fwd(x, G, Cy)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return MbstdBwdFn.apply(x, gy, ctx.G), None, None


class MbstdBwdFn(Function):
    @staticmethod
    def forward(ctx, x, gy, G):
        ctx.G = G
        ctx.save_for_backward(x, gy)
        return ops.mbstd_bwd(x, gy, G)

    @staticmethod
    def backward(ctx, ggx):
        x, gy = ctx.saved_tensors
        G, C = ctx.G, x.shape[1]
        B, M = x.shape[0], x.shape[0] // ctx.G
        with torch.enable_grad():
            xd = x.detach().requires_grad_(True)
            gstat = gy[:, C].reshape(G, M, -1).sum(dim=(0, 2)).detach().requires_grad_(True)  # [M]
            stat = _mbstd_torch(xd, G)
            (gx_stat,) = torch.autograd.grad(stat, xd, gstat, create_graph=True)
            # gx = gy[:, :C] + gx_stat(x, gstat);  contract with ggx
            scalar = (gx_stat * ggx).sum()
            gxd, ggstat = torch.autograd.grad(scalar, (xd, gstat))
        ggy = torch.zeros_like(gy)
        ggy[:, :C] = ggx
        ggy[:, C] = ggstat.repeat(G)[:, None, None].expand(B, *gy.shape[2:])
        return gxd.contiguous(memory_format=ops.CL), ggy, None


def minibatch_stddev(x, group, Cy):
    if x.dtype != torch.float32:  # f16 activations: the statistic of the tiny [B,512,4,4] tensor is taken in fp32
        return MbstdFn.apply(x.float(), group, Cy).to(x.dtype)
    return MbstdFn.apply(x, group, Cy)


# --------------------------------------------------------------------------------------------------------
# generator: fused modulated convolution
# --------------------------------------------------------------------------------------------------------
def _modconv_composite(x, w, s, d, spec, transposed, out_hw, wscale, residual=None, bias=None, act=None):
    """The modulated convolution written with the any-order Functions only (Conv2dFn, BiasActFn, torch broadcasting):
    act(d * conv(s * x, w) + residual + bias).  Same algebra as the fused kernels; used to BUILD A GRAPH of the backward
    when a double backward is requested."""
    z = Conv2dFn.apply(x * s.to(x.dtype)[:, :, None, None], w, spec, transposed, out_hw, wscale, None)
    if d is not None:
        z = z * d.to(z.dtype)[:, :, None, None]
    if act is None:
        return z
    slope, gain = act
    return BiasActFn.apply(z, bias, residual, slope, gain)


def _recorded_backward(inputs, needs, gy, build):
    """Gradients of build(*inputs) w.r.t. the inputs flagged in `needs`, contracted with gy, WITH history (create_graph):
    the forward is re-run through differentiable Functions inside the backward pass.

    The inputs are graph-connected to each other (the demodulation d is a function of the modulation s, s of the style w,
    ...).  torch.autograd.grad(y, [s, d]) would therefore return the TOTAL derivative w.r.t. s, including the path through
    d — which the outer graph then adds a second time when it propagates the returned d-gradient back to s.  Each input is
    replaced by an identity view first: the views are distinct graph nodes with no edges between each other, so every
    returned gradient is the PARTIAL derivative (what a Function's backward must return), while still being connected to
    the original tensors for the next differentiation."""
    with torch.enable_grad():
        proxies = [None if t is None else t.view_as(t) for t in inputs]
        y = build(*proxies)
        idx = [i for i, (t, n) in enumerate(zip(inputs, needs)) if n and t is not None and t.requires_grad]
        grads = torch.autograd.grad(y, [proxies[i] for i in idx], gy, create_graph=True, allow_unused=True)
    out = [None] * len(inputs)
    for i, g in zip(idx, grads):
        out[i] = g
    return out


def _modconv_dgrad_fused(g, w, spec, tr, x, s, d, ws, in_port, want_gs):
    """Gradient of y = d * conv(s * x, w) w.r.t. x (and the modulation gradient gs = sum_hw dxs * x) in ONE launch: the adjoint
    convolution of d * g with the modulation s as its output scale, gs as a dot product against x and — with in_port — the
    backward of the leaky ReLU that produced x, all in the kernel's epilogue (ops.GradFuse).  Replaces data gradient -> dxs,
    mul_reduce(dxs, x) -> (gs, s * dxs) [three full-resolution tensors] and the activation's own backward pass [three more]."""
    slope, gain, want_b, parts = in_port if in_port is not None else (1.0, 1.0, False, None)
    fuse = ops.GradFuse(mask_src=x if in_port is not None else None, mask_slope=slope, mask_gain=gain, want_colsum=want_b,
                        dot_src=x if want_gs else None)
    if not tr:
        gx = ops.conv_bwd_data(g, w, spec, tuple(x.shape[2:]), ws, in_scale=d, out_scale=s, fuse=fuse)
    else:
        gx = ops.conv_fwd(g, w, spec, ws, in_scale=d, out_scale=s, fuse=fuse)
    if want_b:
        parts.append(fuse.colsum)
    return gx, fuse.dot


class ModConvFn(Function):
    """y = d * conv(s * x, w * wscale) with per-sample s [B,Cin] (modulation) and d [B,Cout] (demodulation, or None).
    transposed=True runs the stride-2 up-sampling branch (conv_transpose2d, stylegan2_common_layers.py:322-330).
    in_port: x is the output of a fused leaky ReLU (see ActPort)."""

    @staticmethod
    def forward(ctx, x, w, s, d, spec, transposed, out_hw, wscale, in_port=None, x_port=None, out_f32=False):
        x_in, s_in, d_in = x, s, d  # saved as given (a layout copy made in here would cut the graph of a recorded backward)
        ctx.pmask = _param_mask(x, w, s, d)
        x = ops.nhwc(x)
        s = s.contiguous()
        d = None if d is None else d.contiguous()
        ctx.v = None
        if not transposed:
            # out_f32: f16 activations in, fp32 result out (ToRGB: the RGB skip sum stays fp32 in f16-activation mode)
            y, v = ops.conv_fwd(x, w, spec, wscale, keep_v=True, in_scale=s, out_scale=d, out_f32=out_f32)
            ctx.v = v if ctx.needs_input_grad[1] else None
        else:
            y = ops.conv_bwd_data(x, w, spec, tuple(out_hw), wscale, in_scale=s, out_scale=d)
        ctx.spec, ctx.transposed, ctx.wscale = spec, transposed, wscale
        ctx.out_hw = tuple(y.shape[2:])
        ctx.in_port = in_port
        none = x.new_empty(())  # placeholder for absent tensors (never read): no fill kernel
        ctx.save_for_backward(x_in, w, s_in, d_in if d is not None else none, y if d is not None else none)
        ctx.has_d = d is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, s, d, y = ctx.saved_tensors
        d = d if ctx.has_d else None
        spec, tr, ws = ctx.spec, ctx.transposed, ctx.wscale
        if torch.is_grad_enabled():  # create_graph=True: the gradients must carry history
            gx, gw, gs, gd = _recorded_backward(
                (x, w, s, d), tuple(_need(ctx, i) for i in range(4)), gy,
                lambda x_, w_, s_, d_: _modconv_composite(x_, w_, s_, d_, spec, tr, ctx.out_hw, ws))
            return gx, gw, gs, gd, None, None, None, None, None, None, None
        port = ctx.in_port is not None
        if gy.dtype != x.dtype:  # fp32 result of an f16 layer (out_f32): its gradient re-enters the f16 kernels as f16
            gy = gy.to(x.dtype)
        x, s, gy = ops.nhwc(x), s.contiguous(), ops.nhwc(gy)
        d = None if d is None else d.contiguous()
        O, I = w.shape[:2]
        gx = gs = gd = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            if ops.dot_fusable(x.shape[2], x.shape[3], x.dtype):
                gx, gs = _modconv_dgrad_fused(gy, w, spec, tr, x, s, d, ws, ctx.in_port, ctx.needs_input_grad[2])
            else:
                # gradient w.r.t. (s*x): adjoint conv applied to d*gy (d enters as the input scale of the adjoint)
                if not tr:
                    dxs = ops.conv_bwd_data(gy, w, spec, tuple(x.shape[2:]), ws, in_scale=d)
                else:
                    dxs = ops.conv_fwd(gy, w, spec, ws, in_scale=d)
                gs, gx = ops.mul_reduce(dxs, x, scale=s, want_scaled=True)
                if ctx.in_port is not None:
                    gx = _mask_into_port(ctx.in_port, gx, x)
        if _need(ctx, 1):
            if not tr:
                gw = ops.conv_wgrad(gy, x, spec, O, I, ws, small_scale=d, big_scale=s, big_v=ctx.v)
            else:
                gw = ops.conv_wgrad(x, gy, spec, O, I, ws, small_scale=s, big_scale=d)
        if ctx.has_d and ctx.needs_input_grad[3]:
            num, _ = ops.mul_reduce(gy, y)
            gd = num / d
        return (None if port else gx), gw, gs, gd, None, None, None, None, None, (gx if port else None), None


class ModConvActFn(Function):
    """y = gain*lrelu(d * conv(s * x, w*wscale) + residual + bias): the whole StyledConv (same-resolution branch,
    stylegan2_common_layers.py:479-486) as ONE MFMA kernel launch — modulation on the A-tile load, demodulation,
    condition-noise add, bias and leaky ReLU in the epilogue.  First-order backward = raw fused launches; a recorded
    backward (create_graph=True) differentiates _modconv_composite instead.  Outputs (y, port alias of y); in_port: x is itself
    the output of a fused leaky ReLU (see ActPort)."""

    @staticmethod
    def forward(ctx, x, w, s, d, residual, bias, spec, wscale, slope, gain, in_port=None, parts=None, x_port=None, r_port=None,
                r_alias=None):
        ctx.set_materialize_grads(False)
        ctx.pmask = _param_mask(x, w, s, d, residual, bias)
        saved_in = (x, s, d, residual)  # saved as given (see ModConvFn.forward)
        x = ops.nhwc(x)
        s, d = s.contiguous(), d.contiguous()
        residual = None if residual is None else ops.nhwc(residual)
        y, v = ops.conv_fwd(x, w, spec, wscale, keep_v=True, in_scale=s, out_scale=d, residual=residual, bias=bias, act=True,
                            slope=slope, gain=gain)
        ctx.v = v if ctx.needs_input_grad[1] else None  # Winograd-transformed s*x, reused by the weight gradient
        ctx.cfg = (spec, wscale, slope, gain)
        ctx.has = (residual is not None, bias is not None)
        ctx.in_port, ctx.parts, ctx.r_port = in_port, parts, r_port
        z = x.new_empty(())  # placeholder for absent tensors (never read): no fill kernel
        x_in, s_in, d_in, r_in = saved_in
        ctx.save_for_backward(x_in, w, s_in, d_in, y, r_in if residual is not None else z, bias if bias is not None else z)
        return y, y.detach()  # (not a view of y: see ConvBiasActFn.forward)

    @staticmethod
    def backward(ctx, gy, g_port=None):
        x, w, s, d, y, residual, bias = ctx.saved_tensors
        spec, ws, slope, gain = ctx.cfg
        has_res, has_bias = ctx.has
        residual = residual if has_res else None
        bias = bias if has_bias else None
        O, I = w.shape[:2]
        none9 = (None,) * 9
        if torch.is_grad_enabled():  # create_graph=True: the gradients must carry history (ports carry first-order gradients only)
            assert g_port is None, "an activation port received a gradient inside a recorded backward"
            gx, gw, gs, gd, gr, gb = _recorded_backward(
                (x, w, s, d, residual, bias), tuple(_need(ctx, i) for i in range(6)), gy,
                lambda x_, w_, s_, d_, r_, b_: _modconv_composite(x_, w_, s_, d_, spec, False, None, ws, r_, b_, (slope, gain)))
            return (gx, gw, gs, gd, gr, gb) + none9
        port = ctx.in_port is not None
        x, s, d = ops.nhwc(x), s.contiguous(), d.contiguous()
        residual = None if residual is None else ops.nhwc(residual)
        want_b = has_bias and _need(ctx, 5)
        gpre, gb = _producer_gpre(gy, g_port, y, want_b, slope, gain, ctx.parts)
        gx = gs = gd = gw = None
        if gpre is None:
            return (None,) * 15
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            if ops.dot_fusable(x.shape[2], x.shape[3], x.dtype):
                gx, gs = _modconv_dgrad_fused(gpre, w, spec, False, x, s, d, ws, ctx.in_port, ctx.needs_input_grad[2])
            else:
                dxs = ops.conv_bwd_data(gpre, w, spec, tuple(x.shape[2:]), ws, in_scale=d)
                gs, gx = ops.mul_reduce(dxs, x, scale=s, want_scaled=True)
                if ctx.in_port is not None:
                    gx = _mask_into_port(ctx.in_port, gx, x)
        if _need(ctx, 1):
            gw = ops.conv_wgrad(gpre, x, spec, O, I, ws, small_scale=d, big_scale=s, big_v=ctx.v)
        if ctx.needs_input_grad[3]:
            # d * z = act^-1(y) - residual - bias  =>  gd = sum_hw gpre * z
            gd = ops.act_inv_mul_reduce(gpre, y, residual, bias, slope, gain) / d
        gr = gra = None
        if has_res and ctx.needs_input_grad[4]:
            if ctx.r_port is not None:
                gra = _deliver_residual(ctx.r_port, gpre, gb)
            else:
                gr = gpre
        return ((None if port else gx), gw, gs, gd, gr, gb if want_b else None) + (None,) * 6 + (gx if port else None, None, gra)


def modulated_conv2d_act(x, w, s, d, residual, bias, pad, wscale=1.0, slope=0.2, gain=2 ** 0.5):
    x_port, in_port = _take_port(x)
    port = _new_port(True, slope, gain, bias)
    r_alias, r_port = (None, None) if residual is None else _take_port(residual, identity=True)
    y, alias = ModConvActFn.apply(x, w, s, d, residual, bias, ConvSpec(w.shape[2], w.shape[3], 1, pad), wscale, slope, gain,
                                  in_port, None if port is None else port.parts, x_port, r_port, r_alias)
    return _tag(y, alias, port)


def modulated_conv2d(x, w, s, d, stride=1, pad=0, transposed=False, out_hw=None, wscale=1.0, out_f32=False):
    x_port, in_port = _take_port(x)
    return ModConvFn.apply(x, w, s, d, ConvSpec(w.shape[2], w.shape[3], stride, pad), transposed, out_hw, wscale, in_port, x_port,
                           bool(out_f32) and x.dtype == torch.float16)
