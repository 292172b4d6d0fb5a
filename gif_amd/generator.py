"""MI355X-native GIF generator — drop-in for /root/reference/model/stg2_generator.py:21-333.

Same classes (StyledGenerator, Generator, StyledConvStyleGAN2, ImgEmbedding, ConstantInput), constructor and
forward() signatures, attribute names and state_dict keys (241 keys incl. the double-registered embedding
buffer `image_embedding.embd_weight` / `img_embdng.embd_weight`).  The synthesis network runs on the HIP
kernels through gif_amd.layers (incl. the bilinear condition pyramid); only [B,512]-sized vector math (mapping
network, style scales) stays in torch.  FlameTextureSpace (reference :336-421) is out of scope (needs the
missing photometric_optimization submodule + licensed FLAME assets).
"""
import random

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import functional as GF
from . import ops
from .data import resize_image
from .layers import StyledConv, ToRGB, get_w_frm_z, modulation_bank


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4, constant_background=False):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class ImgEmbedding(nn.Module):
    """Frozen random per-image code book: a BUFFER, not a parameter (reference :34-46)."""

    def __init__(self, vector_size, vocab_size=70_000):
        super().__init__()
        self.register_buffer('embd_weight', torch.randn((vocab_size, vector_size)))

    def get_embddings(self):
        return self.embd_weight

    def forward(self, input):
        return self.embd_weight[input]


class StyledConvStyleGAN2(nn.Module):
    def __init__(self, in_chnl, out_chnl, ker_sz, blur_kernel, noise_in_dims, one_conv_block=False,
                 apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        self.one_conv_block = one_conv_block
        self.st_cv1 = StyledConv(in_chnl, out_chnl, ker_sz, upsample=not one_conv_block, blur_kernel=blur_kernel,
                                 noise_in_dims=noise_in_dims, apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
        if not one_conv_block:
            self.st_cv2 = StyledConv(out_chnl, out_chnl, ker_sz, upsample=False, blur_kernel=blur_kernel,
                                     noise_in_dims=noise_in_dims, apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)

    def forward(self, input, style, noise=None):
        out = self.st_cv1(input, style, noise)
        return out if self.one_conv_block else self.st_cv2(out, style, noise)


# (in, out) channels of the 9 progression blocks, 4x4 ... 1024x1024 (reference :86-114, channel_multiplier=2)
def _block_channels(channel_multiplier):
    cm = channel_multiplier
    return [(512, 512), (512, 512), (512, 512), (512, 512), (512, 256 * cm), (256 * cm, 128 * cm),
            (128 * cm, 64 * cm), (64 * cm, 32 * cm), (32 * cm, 16 * cm)]


class Generator(nn.Module):
    def __init__(self, code_dim, core_tensor_res=4, channel_multiplier=2, noise_in_dims=None,
                 apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        assert core_tensor_res < 64
        assert code_dim == 512
        self.start_step = int(np.log2(core_tensor_res)) - 2
        self.const_input = ConstantInput(512, size=core_tensor_res)
        blur_kernel = [1, 3, 3, 1]
        chans = _block_channels(channel_multiplier)
        self.progression = nn.ModuleList([
            StyledConvStyleGAN2(cin, cout, 3, blur_kernel=blur_kernel, one_conv_block=(i == 0),
                                noise_in_dims=noise_in_dims, apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
            for i, (cin, cout) in enumerate(chans)])
        self.to_rgb = nn.ModuleList([
            ToRGB(cout, code_dim, upsample=(i != 0), apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
            for i, (_, cout) in enumerate(chans)])

    def forward(self, style, pose, noise, step=0, alpha=-1, input_indices=None, mixing_range=(-1, -1)):
        if pose is None:
            out = torch.zeros((noise[0].shape[0], 3), device=noise[0].device)
        else:
            out = pose
        if len(style) < 2:
            inject_index = [len(self.progression) + 1]
        else:
            inject_index = random.sample(list(range(step)), len(style) - 1)
        crossover = 0
        rgb = None
        convs = []
        if len(style) < 2 and mixing_range == (-1, -1):
            # one w for every layer: all modulation linears of this pass in one launch (layers.modulation_bank)
            for i in range(self.start_step, min(step, len(self.progression) - 1) + 1):
                blk = self.progression[i]
                convs += [blk.st_cv1.conv] + ([] if blk.one_conv_block else [blk.st_cv2.conv]) + [self.to_rgb[i].conv]
            modulation_bank(convs, style[0])
        try:
            rgb = self._synthesise(style, out, noise, step, inject_index, crossover, rgb, mixing_range)
        finally:
            # the banked (style, s) pairs hold autograd-graph tensors: a forward that raises or leaves early must not keep them (and the
            # activations behind them) alive until the next forward (advisor finding, round 4)
            for c in convs:
                c._banked = None
        # internal RGB carries zero padding channels in NHWC; hand back the reference's fp32 [B,3,R,R] NCHW tensor
        return [rgb[:, :3].float().contiguous()]

    def _synthesise(self, style, out, noise, step, inject_index, crossover, rgb, mixing_range):
        for i in range(self.start_step, len(self.progression)):
            if mixing_range == (-1, -1):
                if crossover < len(inject_index) and i > inject_index[crossover]:
                    crossover = min(crossover + 1, len(style))
                style_step = style[crossover]
            else:
                style_step = style[1] if mixing_range[0] <= i <= mixing_range[1] else style[0]
            if i == self.start_step:
                out = self.const_input(out).to(noise[i].dtype)  # activation dtype = dtype of the condition pyramid
            out = self.progression[i](out, style_step, noise[i])
            rgb = self.to_rgb[i](out, style_step, rgb)
            if i == step:
                break
        return rgb


class StyledGenerator(nn.Module):
    def __init__(self, n_mlp=8, embedding_vocab_size=1, rendered_flame_ascondition=False, normal_maps_as_cond=False,
                 core_tensor_res=4, w_truncation_factor=1.0, apply_sqrt2_fac_in_eq_lin=False):
        super().__init__()
        noise_in_dims = int(rendered_flame_ascondition * 3 + normal_maps_as_cond * 3)
        self.noise_in_dims = noise_in_dims
        self.core_tensor_res = core_tensor_res
        self.rendered_flame_ascondition = rendered_flame_ascondition
        self.normal_maps_as_cond = normal_maps_as_cond
        self.w_truncation_factor = w_truncation_factor
        self.mean_w = None
        # dtype of the synthesis network's activations in HBM: torch.float32 (reference) or torch.float16 (BASELINE config 5:
        # f16 activations, fp32 weights / demodulation / accumulation).  set_activation_dtype() switches it.
        self.act_dtype = torch.float32
        code_dim = 512
        self.generator = Generator(code_dim, core_tensor_res=core_tensor_res, noise_in_dims=noise_in_dims,
                                   apply_sqrt2_fac_in_eq_lin=apply_sqrt2_fac_in_eq_lin)
        if embedding_vocab_size > 1:
            self.embedding_vocab_size = embedding_vocab_size
            self.image_embedding = ImgEmbedding(vector_size=code_dim, vocab_size=self.embedding_vocab_size)
            self.img_embdng = self.image_embedding  # second registration => both state_dict keys, like the reference
        self.z_to_w = get_w_frm_z(n_mlp, style_dim=code_dim, lr_mlp=0.01, scale_weight=1.0)

    def get_embddings(self):
        return self.image_embedding.get_embddings()

    def set_activation_dtype(self, dtype):
        if dtype not in (torch.float32, torch.float16):
            raise ValueError(f"activation dtype must be torch.float32 or torch.float16, got {dtype}")
        self.act_dtype = dtype
        return self

    def _condition_pyramid(self, cond, step):
        """noise[i] = bilinear resize of the condition to (4*2^i)^2 (reference :309-314), channel-padded NHWC; built in
        fp32, each level cast to the activation dtype."""
        # channel padding (6 -> 8) + NHWC layout in one pass (round 3: F.pad + .contiguous(), two full-resolution ATen passes)
        cond = GF.pack_nhwc(cond, None, ops.cpad(cond.shape[1], self.act_dtype), torch.float32)
        levels = []
        H, W = cond.shape[2:]
        for i in range(step + 1):
            size = 4 * 2 ** i
            if H == W and H == size:
                levels.append(cond)  # the level at the condition's own resolution is the packed tensor itself
            elif H == W and H % size == 0 and (H // size) % 2 == 0:
                levels.append(GF.bilinear_down(cond, size))  # integer ratio: the taps are exact 0.5/0.5 (NHWC kernel)
            else:  # arbitrary ratio (e.g. a 2x2 dummy condition, non-square renders): the generic HIP resampler (csrc/resize.hip)
                lvl = resize_image(cond, (size, size), 'bilinear')
                levels.append(lvl.contiguous(memory_format=torch.channels_last))
        if self.act_dtype != torch.float32:
            levels = [lvl.to(self.act_dtype) for lvl in levels]
        return levels

    def forward(self, input, pose=None, noise=None, step=9, alpha=1, mean_style=None, style_weight=0,
                input_indices=None, mixing_range=(-1, -1)):
        assert step > np.log2(self.core_tensor_res) - 2
        styles = []
        if type(input) not in (list, tuple):
            input = [input]
        if self.rendered_flame_ascondition or self.normal_maps_as_cond:
            if input_indices is None:
                input_indices = torch.zeros(input[0].shape[0], dtype=torch.long, device=input[0].device)
            if input_indices.dtype == torch.float32:  # caller feeds z directly
                styles.append(self.z_to_w(input_indices))
            else:
                w = self.z_to_w(self.img_embdng(input_indices))
                if np.abs(self.w_truncation_factor - 1.0) > 0.01:
                    if self.mean_w is None:
                        self.mean_w = torch.mean(self.z_to_w(self.get_embddings()), dim=0)
                    styles.append(w + (self.mean_w - w) * (1.0 - self.w_truncation_factor))
                else:
                    styles.append(w)
        else:
            for inp in input:
                if self.embedding_vocab_size > 1:
                    if input_indices.dtype == torch.float32:
                        styles.append(torch.cat([inp, input_indices], dim=1))
                    else:
                        styles.append(torch.cat([inp, self.img_embdng(input_indices)], dim=1))
                else:
                    styles.append(inp)
        batch = input[0].shape[0]
        if self.rendered_flame_ascondition or self.normal_maps_as_cond:
            noise = self._condition_pyramid(input[0], step)
        elif noise is None:
            noise = [torch.zeros(batch, ops.cpad(1, self.act_dtype), 4 * 2 ** i, 4 * 2 ** i, device=input[0].device,
                                 dtype=self.act_dtype) for i in range(step + 1)]
        if mean_style is not None:
            styles = [mean_style + style_weight * (style - mean_style) for style in styles]
        return self.generator(styles, pose, noise, step, alpha, input_indices=input_indices, mixing_range=mixing_range)
