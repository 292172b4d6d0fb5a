"""MI355X-native layer library behind the reference's layer API.

Host-side mirror of /root/reference/model/stylegan2_common_layers.py: same class names, constructor
arguments, parameter / buffer names and shapes (so reference state_dicts load with strict=True) and the
same forward() semantics — but every tensor op heavier than a 512x512 linear runs in the hand-written
gfx950 kernels of libgif_hip.so (gif_amd/csrc) through gif_amd.functional.  Internally activations are
NHWC fp32 with channel counts padded to a multiple of 4; module inputs/outputs keep the reference's logical
[B,C,H,W] shapes.  There is no eager/CPU fallback: CPU tensors raise.
"""
import math
import os

import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from . import functional as GF
from .ops import cpad, pad4


_SKINNY_MAX_ROWS = int(_lib.knob("GIF_SKINNY_MAX_ROWS", "512"))  # above: the implicit-GEMM conv kernels take over


def _pad_vec(v, n):
    """flat per-channel vector padded with zeros to n entries (for 3->4 channel RGB tensors)."""
    v = v.reshape(-1)
    k = n - v.numel()
    if k == 0:
        return v
    # cat with a cached zero tail: one launch forward, a view backward (F.pad: fill + copy forward, a copy backward)
    key = (v.device, v.dtype, k)
    tail = _ZERO_TAILS.get(key)
    if tail is None:
        tail = _ZERO_TAILS[key] = torch.zeros(k, device=v.device, dtype=v.dtype)
    return torch.cat([v, tail])


_ZERO_TAILS = {}


class FusedLeakyReLU(nn.Module):
    """sqrt(2) * leaky_relu(x + bias, 0.2) — reference :22-39 — one kernel (gif_bias_act_f32)."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, channel, 1, 1))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input, residual=None):
        return GF.bias_act(input, _pad_vec(self.bias, input.shape[1]), residual, self.negative_slope, self.scale)


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """FIR resampling — reference :42-72 — gif_upfirdn2d_f32."""
    return GF.upfirdn2d(input, kernel, up=up, down=down, pad=pad)


class PixelNorm(nn.Module):
    def forward(self, input):  # [B,512] vectors: plain torch (reference :75-80)
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer('kernel', kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """conv2d(x, W / sqrt(fan_in)) — reference :155-190 — fp32 MFMA implicit GEMM (gif_conv2d_fwd_f32)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        out = GF.conv2d(input, self.weight, self.stride, self.padding, wscale=self.scale)
        if self.bias is not None:
            out = GF.bias_act(out, _pad_vec(self.bias, out.shape[1]), None, 1.0, 1.0)
        return out

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]},'
                f' {self.weight.shape[2]}, stride={self.stride}, padding={self.padding})')


class EqualLinear(nn.Module):
    """Reference :193-235 — y = [sqrt(2) *] lrelu(x @ (W * scale)^T + bias * lr_mul).

    Batch-sized inputs (the mapping network's 8 layers, every modulation linear, the discriminator head) run on the fp32
    skinny-GEMM kernels of csrc/linear.hip: ONE launch per layer forward with the equalised-lr scale, bias and leaky ReLU
    in the epilogue; inputs with many rows fall through to the implicit-GEMM conv kernels (a 1x1 convolution).  Backward =
    any-order Functions in both cases (R1 differentiates the head twice, the StyleGAN2-form path-length regulariser the
    mapping network).  No CPU path."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None, scale_weight=1.0,
                 apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul / scale_weight))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul
        self.apply_sqrt2_fac_in_eq_lin = apply_sqrt2_fac_in_eq_lin

    def forward(self, input):
        out_dim, in_dim = self.weight.shape
        lead = input.shape[:-1]
        x = input.reshape(-1, in_dim)
        if x.dtype != torch.float32:
            x = x.float()
        np_ = pad4(out_dim)
        # (lr_mul == 1 for every modulation linear: no multiply launch)
        bias = None if self.bias is None else _pad_vec(self.bias if self.lr_mul == 1 else self.bias * self.lr_mul, np_)
        gain = (1.41421356237 if self.apply_sqrt2_fac_in_eq_lin else 1.0) if self.activation else 1.0
        if in_dim % 4 == 0 and x.shape[0] <= _SKINNY_MAX_ROWS and np_ <= 1024:
            # batch-sized row counts: the skinny-GEMM kernels (csrc/linear.hip), one launch, epilogue fused
            out = GF.linear_bias_act(x, self.weight, bias, self.scale, bool(self.activation), 0.2, gain, np_)
        else:
            # many rows (e.g. the mean style over the whole code book) or an odd width: a 1x1 convolution over a
            # [rows, in_dim, 1, 1] image on the implicit-GEMM kernels (weight scale folded into the cached packing)
            kp = pad4(in_dim)
            if kp != in_dim:
                x = F.pad(x, (0, kp - in_dim))
            x = x.reshape(x.shape[0], kp, 1, 1)
            w = self.weight.view(out_dim, in_dim, 1, 1)
            if self.activation:
                out = GF.conv2d_bias_act(x, w, bias, 1, 0, self.scale, 0.2, gain)
            elif bias is not None:
                out = GF.conv2d_bias_act(x, w, bias, 1, 0, self.scale, 1.0, 1.0)  # slope 1 / gain 1: bias only
            else:
                out = GF.conv2d(x, w, 1, 0, wscale=self.scale)
            out = out.reshape(out.shape[0], -1)
        if out.shape[1] != out_dim:
            out = out[:, :out_dim]
        return out.reshape(*lead, out_dim)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return GF.bias_act(input, None, None, self.negative_slope, math.sqrt(2))


class ModulatedConv2d(nn.Module):
    """Reference :250-349.  The per-sample weight tensor of the reference (groups=batch) is never built:
    y = d[b,co] * conv(s[b,ci] * x, scale*W), d = rsqrt(scale^2 * sum_ci s^2 * sum_k W^2 + eps) — identical algebra,
    run as one fused MFMA kernel (GF.modulated_conv2d)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1], apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1,
                                      apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
        self.demodulate = demodulate
        self.out_fp32 = False  # f16-activation mode: hand the result back as fp32 (set by ToRGB)

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, '
                f'upsample={self.upsample}, downsample={self.downsample})')

    _banked = None  # (style tensor, s) handed over by modulation_bank for the next call of scales()

    def scales(self, style, in_act=None, out_act=None):
        """(s [B, in_act], d [B, out_act] or None) in fp32, padded to the activations' channel counts (padded s lanes meet zero
        activations, padded d lanes are 1).  s: one skinny GEMM (EqualLinear); d: one more with the square / rsqrt in its operand
        load and epilogue (GF.demodulation)."""
        in_act = self.in_channel if in_act is None else in_act
        out_act = self.out_channel if out_act is None else out_act
        mod = self.modulation
        x = style.reshape(-1, style.shape[-1])
        if x.dtype != torch.float32:
            x = x.float()
        fast = x.shape[0] <= _SKINNY_MAX_ROWS and self.in_channel % 4 == 0 and in_act <= 1024 and mod.weight.shape[1] % 4 == 0
        banked, self._banked = self._banked, None
        if banked is not None and banked[0] is style and in_act == self.in_channel and fast:
            s = banked[1]  # computed with every other layer's modulation in one launch (modulation_bank below)
        elif fast:
            bias = None if mod.bias is None else _pad_vec(mod.bias if mod.lr_mul == 1 else mod.bias * mod.lr_mul, in_act)
            s = GF.linear_bias_act(x, mod.weight, bias, mod.scale, False, 0.2, 1.0, in_act)  # columns >= in_channel: zero
        else:
            s = mod(style)
            if in_act != self.in_channel:
                s = F.pad(s, (0, in_act - self.in_channel), value=1.0)
        d = None
        if self.demodulate:
            if fast and out_act <= 1024:
                d = GF.demodulation(s, self.weight.squeeze(0), self.scale, self.eps, out_act)
            else:
                wsq = self.weight.squeeze(0).pow(2).sum(dim=(2, 3))  # [Cout, Cin]
                cin, cout = self.in_channel, wsq.shape[0]
                s2 = s[:, :cin].pow(2)
                if pad4(cin) != cin:
                    s2 = F.pad(s2, (0, pad4(cin) - cin))
                acc = GF.conv2d(s2.reshape(s.shape[0], -1, 1, 1), wsq.view(cout, cin, 1, 1), 1, 0, wscale=self.scale ** 2)
                d = torch.rsqrt(acc.reshape(s.shape[0], -1)[:, :cout] + self.eps)
                if out_act != self.out_channel:
                    d = F.pad(d, (0, out_act - self.out_channel), value=1.0)
        return s, d

    def _padded_scales(self, style, in_act, dtype=torch.float32):
        """fp32 scales padded to the activation's channel counts (multiples of 4 for fp32, 8 for f16 activations)."""
        return self.scales(style, in_act, cpad(self.out_channel, dtype))

    def forward_fused_act(self, input, style, residual, bias, slope=0.2, gain=2 ** 0.5):
        """act(modconv(input, style) + residual + bias) with everything after the contraction fused into the kernel
        epilogues: same-resolution branch = one MFMA launch; up-sampling branch = conv_transpose + one FIR launch."""
        batch, in_act, height, width = input.shape
        s, d = self._padded_scales(style, in_act, input.dtype)
        w = self.weight.squeeze(0)
        if self.upsample:
            out = GF.modulated_conv2d(input, w.transpose(0, 1), s, d, stride=2, pad=0, transposed=True,
                                      out_hw=(2 * height + self.kernel_size - 2, 2 * width + self.kernel_size - 2),
                                      wscale=self.scale)
            return GF.blur_bias_act(out, self.blur.kernel, self.blur.pad, residual, bias, slope, gain)
        if self.downsample or d is None:  # (the scales are already computed — possibly by the modulation bank: no second launch)
            return GF.bias_act(self.forward(input, style, scales=(s, d)), bias, residual, slope, gain)
        return GF.modulated_conv2d_act(input, w, s, d, residual, bias, self.padding, self.scale, slope, gain)

    def forward(self, input, style, scales=None):
        batch, in_act, height, width = input.shape
        s, d = self._padded_scales(style, in_act, input.dtype) if scales is None else scales
        w = self.weight.squeeze(0)  # [Cout, Cin, k, k]
        if self.upsample:
            # conv_transpose2d(x, W^T, stride 2): underlying forward conv maps Cout -> Cin, so canonical = W^T view
            out = GF.modulated_conv2d(input, w.transpose(0, 1), s, d, stride=2, pad=0, transposed=True,
                                      out_hw=(2 * height + self.kernel_size - 2, 2 * width + self.kernel_size - 2),
                                      wscale=self.scale)
            out = self.blur(out)
        elif self.downsample:
            out = GF.modulated_conv2d(self.blur(input), w, s, d, stride=2, pad=0, wscale=self.scale)
        else:
            out = GF.modulated_conv2d(input, w, s, d, stride=1, pad=self.padding, wscale=self.scale, out_f32=self.out_fp32)
        return out


_STYLE_BANK = _lib.knob("GIF_STYLE_BANK", "1") != "0"


def modulation_bank(convs, style):
    """The modulation linears of `convs` (ModulatedConv2d layers about to be called with the same `style`) as ONE launch
    (GF.linear_bank; reference: one EqualLinear call per layer, stylegan2_common_layers.py:311-313).  Each layer picks its s up in
    its next scales() call; layers the bank does not take (odd widths, a different style tensor) compute their own as before."""
    if not _STYLE_BANK or len(convs) < 2:
        return
    x = style.reshape(-1, style.shape[-1])
    mods = [c.modulation for c in convs]
    m0 = mods[0]
    same = all(m.scale == m0.scale and m.lr_mul == 1 and m.bias is not None and m.activation is None and
               m.weight.shape[1] == m0.weight.shape[1] for m in mods)
    if not same or x.dtype != torch.float32 or x.shape[0] > _SKINNY_MAX_ROWS or any(c.in_channel > 1024 for c in convs):
        return
    weights = [m.weight for m in mods]
    if not GF.ops.linear_bank_ok(x, weights):
        return
    outs = GF.linear_bank(x, weights, [m.bias for m in mods], m0.scale)
    for c, o in zip(convs, outs):
        c._banked = (style, o)


class NoiseInjection(nn.Module):
    """GIF's condition-driven 'noise': 3x(conv3x3+bias), ReLU between — reference :388-431."""

    @staticmethod
    def small_init_weights(m):
        if hasattr(m, 'weight'):
            m.weight.data = torch.randn_like(m.weight) / 100
        if hasattr(m, 'bias'):
            m.bias.data.fill_(0.0001)

    def __init__(self, noise_in_chalnnels, noise_out_channels):
        super().__init__()
        self.noise_in_chalnnels = noise_in_chalnnels
        c = noise_in_chalnnels
        # nn.Conv2d / nn.ReLU are parameter containers only (state_dict keys noise_conv.{0,2,4}.*); the math is HIP
        self.noise_conv = nn.Sequential(
            nn.Conv2d(c, 2 * c, 3, padding=1), nn.ReLU(),
            nn.Conv2d(2 * c, 4 * c, 3, padding=1), nn.ReLU(),
            nn.Conv2d(4 * c, noise_out_channels, 3, padding=1))
        self.noise_conv.apply(NoiseInjection.small_init_weights)

    def convolve(self, noise):
        h = noise
        for idx in (0, 2, 4):
            conv = self.noise_conv[idx]
            last = idx == 4
            # bias (+ReLU) run in the conv kernel's epilogue: leaky_relu with slope 0 / gain 1 is the ReLU; the last conv
            # has no activation (slope 1 = identity)
            h = GF.conv2d_bias_act(h, conv.weight, _pad_vec(conv.bias, cpad(conv.weight.shape[0], h.dtype)), 1, 1, 1.0,
                                   1.0 if last else 0.0, 1.0)
        return h

    def forward(self, image, noise):
        batch, _, height, width = image.shape
        if noise is None:
            noise = image.new_empty(batch, cpad(self.noise_in_chalnnels, image.dtype), height, width).normal_()
            noise[:, self.noise_in_chalnnels:] = 0
        return GF.bias_act(image, None, self.convolve(noise), 1.0, 1.0)


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """act(modconv(x, w) + noise_conv(cond) + bias) — reference :447-486; the add/bias/lrelu are one kernel."""

    def __init__(self, in_channel, out_channel, kernel_size, noise_in_dims, style_dim=512, upsample=False,
                 blur_kernel=[1, 3, 3, 1], demodulate=True, apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate,
                                    apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
        self.noise = NoiseInjection(noise_in_dims, out_channel)
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None):
        if noise is None:  # random-noise fallback of the reference (:424-425): unfused
            return self.activate(self.noise(self.conv(input, style), noise=None))
        act = self.activate
        return self.conv.forward_fused_act(input, style, self.noise.convolve(noise),
                                           _pad_vec(act.bias, cpad(self.conv.out_channel, input.dtype)), act.negative_slope,
                                           act.scale)


class ToRGB(nn.Module):
    """1x1 modulated conv without demodulation + bias + upsampled skip — reference :489-511.
    Internally RGB tensors carry 4 channels (the 4th is zero)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1],
                 apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False,
                                    apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
        # f16 activations: the running RGB image (|v| up to ~8, where half resolves 3.9e-3) is accumulated in fp32 — it is three
        # channels; the feature maps stay f16 (profiles/r3_f16_error_by_layer.txt)
        self.conv.out_fp32 = True
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        out = self.conv(input, style)
        if skip is not None:
            skip = self.upsample(skip)
        return GF.bias_act(out, _pad_vec(self.bias, out.shape[1]), skip, 1.0, 1.0)


def get_w_frm_z(n_mlp, style_dim, lr_mlp=1, scale_weight=1.0):
    if n_mlp > 0:
        layers = [PixelNorm()]
        for i in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu',
                                      scale_weight=scale_weight))
        return nn.Sequential(*layers)

    class Net(nn.Module):
        def forward(self, *args):
            return args[0]

    return Net()


class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> FusedLeakyReLU — reference :752-799."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride = 2
            self.padding = 0
        else:
            stride = 1
            self.padding = kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)

    def forward(self, input, passthrough=False, out_scale=1.0, residual=None):
        """passthrough=True (fused, non-downsampling layers only): returns (out, input') with input' an alias of the
        input for a second consumer — see functional.ConvBiasActFn.
        out_scale / residual (ResBlock): the layer's result is multiplied by out_scale (folded into the activation gain or
        the weight scale — no extra pass) and, for the blur + 1x1 layer, `residual` is added in the conv epilogue."""
        mods = list(self)
        if isinstance(mods[0], Blur):
            assert not passthrough
            nxt = mods[1]
            if (isinstance(nxt, EqualConv2d) and nxt.weight.shape[2] == 1 and nxt.weight.shape[3] == 1 and nxt.stride == 2
                    and nxt.padding == 0 and len(mods) == 2 and nxt.bias is None):
                # blur followed by a 1x1 stride-2 conv (the ResBlock skip): the conv only ever reads every second blurred
                # pixel, so blur AND decimate in one FIR pass (a quarter of the outputs) and run the 1x1 conv at stride 1 —
                # the same taps for the kept pixels, i.e. the same function, without the full-resolution blurred tensor
                input = upfirdn2d(input, mods[0].kernel, up=1, down=2, pad=mods[0].pad)
                return GF.conv2d(input, nxt.weight, 1, 0, wscale=nxt.scale * out_scale, residual=residual)
            input = mods[0](input)
            mods = mods[1:]
        assert residual is None, "residual is only fused into the blur + 1x1 layer"
        conv = mods[0]
        if len(mods) == 2 and isinstance(mods[1], FusedLeakyReLU) and conv.bias is None:
            act = mods[1]  # EqualConv2d + FusedLeakyReLU => bias and lrelu run in the conv kernel's epilogue
            return GF.conv2d_bias_act(input, conv.weight, _pad_vec(act.bias, cpad(conv.weight.shape[0], input.dtype)), conv.stride,
                                      conv.padding, conv.scale, act.negative_slope, act.scale * out_scale, passthrough)
        assert out_scale == 1.0, "out_scale needs one of the fused layer forms"
        first = input
        for m in mods:
            input = m(input)
        return (input, first) if passthrough else input


class ResBlock(nn.Module):
    """Reference :802-820; (out + skip)/sqrt(2) runs in the conv epilogues (no elementwise pass)."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        # the skip branch consumes the alias handed back by conv1: its input gradient is then accumulated inside conv1's
        # data-gradient kernel instead of by a separate add over two full-resolution tensors
        # (conv2(conv1(x)) + skip(x)) / sqrt(2) without a separate add/scale pass: 1/sqrt(2) is folded into conv2's
        # activation gain and into the skip conv's weight scale, and the skip conv adds conv2's output in its epilogue
        out, input_alias = self.conv1(input, passthrough=True)
        r = 1 / math.sqrt(2)
        out = self.conv2(out, out_scale=r)
        return self.skip(input_alias, out_scale=r, residual=out)
