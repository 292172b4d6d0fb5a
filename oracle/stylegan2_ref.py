"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by gif_amd/).

Plain-PyTorch fp32 restatement of the reference's generator / discriminator hot path as PURE FUNCTIONS of a
state_dict (no nn.Module), so it can travel to the GPU box where /root/reference does not exist.  Each function
cites the reference lines it restates (paths relative to /root/reference/).  It is deliberately naive (per-sample
weights, groups=batch, unfused), i.e. the reference's own algorithm, not the product's.

Parity status: PINNED — tests/test_oracle_stylegan2.py checks every function against the imported reference
modules (same state_dict, same inputs) in the build container, and against golden vectors generated from the
reference (tests/golden/make_stylegan2_golden.py) everywhere else.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = 2 ** 0.5


# ---- model/stylegan2_common_layers.py --------------------------------------------------------------------
def make_kernel(k):  # :83-91
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):  # :42-72
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    z = x.reshape(B * C, 1, H, 1, W, 1)
    z = F.pad(z, (0, up - 1, 0, 0, 0, up - 1))  # zero insertion after every sample
    z = z.reshape(B * C, 1, H * up, W * up)
    z = F.pad(z, (max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)))
    z = z[:, :, max(-p0, 0): z.shape[2] - max(-p1, 0), max(-p0, 0): z.shape[3] - max(-p1, 0)]
    z = F.conv2d(z, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw))  # flipped => true convolution (:64-65)
    z = z.reshape(B, C, z.shape[2], z.shape[3])
    return z[:, :, ::down, ::down]


def fused_leaky_relu(x, bias):  # FusedLeakyReLU :22-39 ; bias [1,C,1,1]
    return SQRT2 * F.leaky_relu(x + bias, 0.2)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):  # EqualLinear :193-235
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return F.leaky_relu(F.linear(x, weight * scale) + bias * lr_mul, 0.2)
    return F.linear(x, weight * scale, bias=None if bias is None else bias * lr_mul)


def equal_conv2d(x, weight, bias=None, stride=1, padding=0):  # EqualConv2d :155-184
    scale = 1 / math.sqrt(weight.shape[1] * weight.shape[2] ** 2)
    return F.conv2d(x, weight * scale, bias=bias, stride=stride, padding=padding)


def pixel_norm(x):  # :75-80
    return x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def modulated_conv2d(x, weight, mod_w, mod_b, style, demodulate=True, upsample=False, blur_k=None, downsample=False):
    """ModulatedConv2d.forward :307-349 (weight [1,Co,Ci,k,k]; modulation = EqualLinear(512->Ci, bias_init 1))."""
    B, Ci, H, W = x.shape
    _, Co, _, k, _ = weight.shape
    s = equal_linear(style, mod_w, mod_b).view(B, 1, Ci, 1, 1)
    w = (1 / math.sqrt(Ci * k * k)) * weight * s
    if demodulate:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(B, Co, 1, 1, 1)
    if upsample:  # :322-333
        wt = w.transpose(1, 2).reshape(B * Ci, Co, k, k)
        out = F.conv_transpose2d(x.reshape(1, B * Ci, H, W), wt, padding=0, stride=2, groups=B)
        out = out.view(B, Co, out.shape[2], out.shape[3])
        p = (4 - 2) - (k - 1)
        return upfirdn2d(out, blur_k, pad=((p + 1) // 2 + 1, p // 2 + 1))  # Blur(pad0,pad1) of :272-278
    if downsample:  # :335-341 (Blur with pad computed at :280-286, then stride-2 grouped conv)
        p = (4 - 2) + (k - 1)
        xb = upfirdn2d(x, blur_k, pad=((p + 1) // 2, p // 2))
        out = F.conv2d(xb.reshape(1, B * Ci, xb.shape[2], xb.shape[3]), w.view(B * Co, Ci, k, k), padding=0, stride=2,
                       groups=B)
        return out.view(B, Co, out.shape[2], out.shape[3])
    out = F.conv2d(x.reshape(1, B * Ci, H, W), w.view(B * Co, Ci, k, k), padding=k // 2, groups=B)  # :343-347
    return out.view(B, Co, out.shape[2], out.shape[3])


def noise_conv(sd, prefix, cond):  # NoiseInjection.noise_conv :405-414
    h = F.relu(F.conv2d(cond, sd[prefix + '0.weight'], sd[prefix + '0.bias'], padding=1))
    h = F.relu(F.conv2d(h, sd[prefix + '2.weight'], sd[prefix + '2.bias'], padding=1))
    return F.conv2d(h, sd[prefix + '4.weight'], sd[prefix + '4.bias'], padding=1)


def styled_conv(sd, prefix, x, style, cond, upsample):  # StyledConv.forward :479-486
    out = modulated_conv2d(x, sd[prefix + 'conv.weight'], sd[prefix + 'conv.modulation.weight'],
                           sd[prefix + 'conv.modulation.bias'], style, True, upsample,
                           sd.get(prefix + 'conv.blur.kernel'))
    out = out + noise_conv(sd, prefix + 'noise.noise_conv.', cond)  # NoiseInjection.forward :421-431
    return fused_leaky_relu(out, sd[prefix + 'activate.bias'])


def to_rgb(sd, prefix, x, style, skip):  # ToRGB.forward :502-511
    out = modulated_conv2d(x, sd[prefix + 'conv.weight'], sd[prefix + 'conv.modulation.weight'],
                           sd[prefix + 'conv.modulation.bias'], style, demodulate=False)
    out = out + sd[prefix + 'bias']
    if skip is not None:
        out = out + upfirdn2d(skip, sd[prefix + 'upsample.kernel'], up=2, pad=(2, 1))  # Upsample :94-112
    return out


# ---- model/stg2_generator.py -----------------------------------------------------------------------------
def z_to_w(sd, z, n_mlp=8, prefix='z_to_w.'):  # get_w_frm_z :514-524, lr_mlp = 0.01 (stg2_generator.py:237)
    h = pixel_norm(z)
    for i in range(1, n_mlp + 1):
        h = equal_linear(h, sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias'], lr_mul=0.01, activation=True)
    return h


def generator_forward(sd, cond, step, input_indices, n_mlp=8):
    """StyledGenerator.forward :249-328 + Generator.forward :159-209 for the rendered-condition configuration
    (rendered_flame_ascondition / normal_maps_as_cond, pose=None, single style, core_tensor_res=4).
    input_indices: int64 [B] (embedding lookup :275) or float32 [B,512] (z fed directly :272-273)."""
    if input_indices.dtype == torch.float32:
        w = z_to_w(sd, input_indices, n_mlp)
    else:
        w = z_to_w(sd, sd['image_embedding.embd_weight'][input_indices], n_mlp)
    B = cond.shape[0]
    out = sd['generator.const_input.input'].repeat(B, 1, 1, 1)  # ConstantInput :27-31
    rgb = None
    for i in range(step + 1):
        size = 4 * 2 ** i
        c_i = F.interpolate(cond, size=(size, size), mode='bilinear', align_corners=False)  # :309-314
        p = f'generator.progression.{i}.'
        out = styled_conv(sd, p + 'st_cv1.', out, w, c_i, upsample=(i != 0))  # StyledConvStyleGAN2 :62-66
        if i != 0:
            out = styled_conv(sd, p + 'st_cv2.', out, w, c_i, upsample=False)
        rgb = to_rgb(sd, f'generator.to_rgb.{i}.', out, w, rgb)
    return rgb


# ---- model/stg2_discriminator.py -------------------------------------------------------------------------
def conv_layer(sd, prefix, x, k, downsample, activate=True):  # ConvLayer :752-799
    idx = 0
    if downsample:
        p = (4 - 2) + (k - 1)
        x = upfirdn2d(x, sd[f'{prefix}{idx}.kernel'], pad=((p + 1) // 2, p // 2))
        idx += 1
    x = equal_conv2d(x, sd[f'{prefix}{idx}.weight'], None, stride=2 if downsample else 1,
                     padding=0 if downsample else k // 2)
    if activate:
        x = fused_leaky_relu(x, sd[f'{prefix}{idx + 1}.bias'])
    return x


def res_block(sd, prefix, x):  # ResBlock :802-820
    out = conv_layer(sd, prefix + 'conv1.', x, 3, False)
    out = conv_layer(sd, prefix + 'conv2.', out, 3, True)
    skip = conv_layer(sd, prefix + 'skip.', x, 1, True, activate=False)
    return (out + skip) / math.sqrt(2)


def minibatch_stddev(out, stddev_group=4):  # stg2_discriminator.py:56-65
    B, C, H, W = out.shape
    group = min(B, stddev_group)
    sd_ = out.view(group, -1, 1, C, H, W)
    sd_ = torch.sqrt(sd_.var(0, unbiased=False) + 1e-8)
    sd_ = sd_.mean([2, 3, 4], keepdim=True).squeeze(2)
    sd_ = sd_.repeat(group, 1, H, W)
    return torch.cat([out, sd_], 1)


def discriminator_forward(sd, image, condition, size):  # Discriminator.forward :48-76
    x = torch.cat((image, condition), dim=1) if condition is not None else image
    out = conv_layer(sd, 'convs.0.', x, 1, False)
    for j in range(1, int(math.log2(size)) - 1):
        out = res_block(sd, f'convs.{j}.', out)
    out = minibatch_stddev(out)
    out = conv_layer(sd, 'final_conv.', out, 3, False)
    out = out.reshape(out.shape[0], -1)
    out = equal_linear(out, sd['final_linear.0.weight'], sd['final_linear.0.bias'], activation=True)
    return equal_linear(out, sd['final_linear.1.weight'], sd['final_linear.1.bias'])


# ---- loss_functions/losses.py ----------------------------------------------------------------------------
def grad_penalty_loss(inputs, outs):  # :87-99 with step=None => weight 5.0 ; returns [B]
    pen = 0
    for inpt in inputs:
        g = torch.autograd.grad(outputs=outs.sum(), inputs=inpt, create_graph=True)[0]
        pen = pen + 5.0 * (g.reshape(g.size(0), -1).norm(2, dim=1) ** 2)
    return pen


# ---- deterministic test weights ----------------------------------------------------------------------------
def seeded_state_dict(template_sd, seed, std_overrides=None):
    """Fill a state_dict of the given key->shape template with reproducible values:
    parameters ~ their reference init distribution scale (so activations stay O(1)), FIR kernels kept as they are."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template_sd.keys()):
        v = template_sd[k]
        if k.endswith('.kernel'):
            out[k] = v.clone().float()
            continue
        t = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if 'noise_conv' in k:
            t = t * (0.05 if k.endswith('weight') else 0.01)
        elif k.endswith('modulation.bias'):
            t = 1.0 + 0.1 * t
        elif k.endswith('.bias') or k.endswith('activate.bias'):
            t = 0.1 * t
        elif k.startswith('z_to_w') and k.endswith('weight'):
            t = t / 0.01  # EqualLinear(lr_mul=0.01) stores randn / lr_mul (:198)
        out[k] = t
    if 'image_embedding.embd_weight' in out:
        out['img_embdng.embd_weight'] = out['image_embedding.embd_weight']
    return out
