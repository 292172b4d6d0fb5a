"""Regularisers of the GIF training step on the HIP path — mirror of loss_functions/losses.py:87-124.

grad_penalty_loss (R1) keeps the reference signature; the per-sample squared norm is a HIP reduction
(gif_sqnorm_per_sample_f32) wrapped so that it stays differentiable (the penalty is back-propagated through D's
backward: every D layer is an any-order autograd Function, gif_amd/functional.py).
"""
import numpy as np
import torch
from torch.autograd import Function, grad
from torch.autograd.function import once_differentiable

from . import ops
from .functional import inputs_only_backward


class _SqNormFn(Function):
    """out[b] = sum(g[b]^2); backward 2*g*gout[b] (itself differentiable through torch ops)."""

    @staticmethod
    def forward(ctx, g):
        ctx.save_for_backward(g)
        return ops.sqnorm_per_sample(g)

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * (2.0 * gout).view(-1, *([1] * (g.dim() - 1)))


def sqnorm_per_sample(g):
    return _SqNormFn.apply(g)


def l2_reg(model):  # losses.py:16-20
    reg = 0
    for param in model.parameters():
        reg = reg + torch.norm(param)
    return reg


def grad_penalty_loss(inputs, outs, step, grad_scale=1.0):
    """R1: sum over inputs of w * ||d sum(outs) / d input||^2 per sample (losses.py:87-99).  Returns [B].
    grad_scale (not in the reference; 1.0 = reference arithmetic): the gradient is taken of grad_scale * sum(outs) and divided
    back — used with f16 activations, whose first-backward activation gradients would otherwise underflow."""
    grad_penalty = 0
    for inp_idx, inpt in enumerate(inputs):
        with inputs_only_backward():  # the inner pass needs d outs / d input only: no weight / bias gradients (functional.py)
            if grad_scale == 1.0:
                grad_real = grad(outputs=outs.sum(), inputs=inpt, create_graph=True)[0]
            else:
                grad_real = grad(outputs=outs.sum() * grad_scale, inputs=inpt, create_graph=True)[0] / grad_scale
        if step is not None:
            w = 1 + step - inp_idx
            w = 0.05 / (w * np.log2(1 + w))
        else:
            w = 5.0
        grad_penalty = grad_penalty + w * sqnorm_per_sample(grad_real)
    return grad_penalty


class PathLengthRegularizor:
    """losses.py:102-124.  reference_semantics=True reproduces the reference arithmetic exactly: whole-batch numel in
    the noise scale, NO create_graph (so the penalty carries no gradient to G), scalar mean length, and the
    "moving mean" update mean + 0.01*len - mean.  reference_semantics=False is the StyleGAN2 form (per-image noise
    scale, create_graph, per-sample lengths, true EMA).  The reference draws a [B,159] FLAME vector as the
    differentiation variable, which only works for a vector-conditioned G; for the rendered-condition G (the only
    shipped configuration, where the reference code cannot run) the variable is the z fed through `input_indices`
    (float32 => z path, stg2_generator.py:272-273) and `cond` is the condition image."""

    def __init__(self, reference_semantics=True):
        self.pl_moving_mean = 0
        self.pl_decay = 0.01
        self.reference_semantics = reference_semantics

    def path_length_reg(self, generator, step, alpha, input_indices, cond=None):
        dev = input_indices.device
        B = input_indices.shape[0]
        if cond is None:
            style = torch.randn((B, 159), device=dev, requires_grad=True)
            fake = generator(input=style, noise=None, step=step, alpha=alpha, input_indices=input_indices)[0]
        else:
            style = torch.randn((B, 512), device=dev, requires_grad=True)
            fake = generator(cond, None, step=step, alpha=alpha, input_indices=style)[0]
        if self.reference_semantics:
            noise = torch.randn(fake.shape, device=dev) / np.sqrt(np.prod(fake.shape))
            with inputs_only_backward():
                pl_grads = grad(outputs=torch.sum(fake * noise), inputs=style)[0]
            pl_lengths = torch.mean(torch.sqrt(sqnorm_per_sample(pl_grads)))
            self.pl_moving_mean = self.pl_moving_mean + self.pl_decay * pl_lengths - self.pl_moving_mean
            return torch.pow(pl_lengths - self.pl_moving_mean, 2)
        noise = torch.randn(fake.shape, device=dev) / np.sqrt(fake.shape[2] * fake.shape[3])
        with inputs_only_backward():
            pl_grads = grad(outputs=torch.sum(fake * noise), inputs=style, create_graph=True)[0]
        pl_lengths = torch.sqrt(sqnorm_per_sample(pl_grads))
        mean = self.pl_moving_mean + self.pl_decay * (pl_lengths.mean().detach() - self.pl_moving_mean)
        self.pl_moving_mean = mean
        return (pl_lengths - mean).pow(2).mean()


# --------------------------------------------------------------------------------------------------------
# texture-interpolation loss (loss_functions/losses.py:127-243) — the FLAME-free core
# --------------------------------------------------------------------------------------------------------
def interpolate_flame_labels(flm_lbls, t=None):
    """train.py:224-227: neighbouring samples' FLAME parameters (shape / expression / pose / camera: the first 159 labels) are
    blended with ONE random weight t ~ U(0,1) for the whole batch (np.random.uniform, like the reference; pass t to replay a
    draw); light and texture codes (labels 159..) stay those of the first sample of each pair.  [B,L] -> [B-1,L]."""
    if t is None:
        t = np.random.uniform(0, 1)
    a = flm_lbls[:-1, :159] + t * (flm_lbls[1:, :159] - flm_lbls[:-1, :159])
    return torch.cat((a, flm_lbls[:-1, 159:]), dim=-1)


class _TexPairLossFn(Function):
    """mean(sigmoid(((a - b) * ma * mb)^2) * f): one fused HIP reduction + one pointwise backward (once differentiable)."""

    @staticmethod
    def forward(ctx, a, b, ma, mb, f):
        ctx.save_for_backward(a, b, f)
        ctx.masks = (ma, mb)
        return ops.texture_pair_loss(a, b, ma, mb, f)

    @staticmethod
    @once_differentiable  # raw kernel launch: a double backward raises instead of returning a history-free gradient
    def backward(ctx, gloss):
        a, b, f = ctx.saved_tensors
        ga = ops.texture_pair_loss(a, b, ctx.masks[0], ctx.masks[1], f, gloss=gloss)
        return (ga if ctx.needs_input_grad[0] else None), (-ga if ctx.needs_input_grad[1] else None), None, None, None


class InterpolatedTextureLoss:
    """Mirror of InterpolatedTextureLoss (loss_functions/losses.py:127-243): generated images of DIFFERENT expressions/poses
    but the same identity must agree in UV texture space wherever both see the surface.

    The reference constructor loads the licensed FLAME texture space, a face-region mask image and builds the FLAME
    renderer (`OverLayViz`) — none of which exist in this repository.  Here they are injected:
      face_region_only_mask  [1,1,h,w] float in [0,1]  (reference: cnst.face_region_mask_file / 255)
      flm_tex_dec            gif_amd.texture_space.FlameTextureSpace (or any callable images, flame_params -> textures, masks)
      render_condition       callable flame_batch -> (rend_flm, norma_map_img) in [-1,1]  (reference: OverLayViz + clamp*2-1)
    `pairwise_texture_loss` and `texture_pairs_loss` (the loop of tex_sp_intrp_loss) run on the fused HIP kernel."""

    def __init__(self, max_images_in_batch, face_region_only_mask, flm_tex_dec=None, render_condition=None):
        self.face_region_only_mask = face_region_only_mask.to(torch.float32)
        self.flm_tex_dec = flm_tex_dec
        self.render_condition = render_condition
        self.max_num = max_images_in_batch - 1
        self.pairs = np.array([(i, j) for i in range(self.max_num) for j in range(i + 1, self.max_num)])

    def _face_mask(self, like):
        from .data import fast_image_reshape
        if self.face_region_only_mask.device != like.device:
            self.face_region_only_mask = self.face_region_only_mask.to(like.device)
        if self.face_region_only_mask.shape[-1] != like.shape[-1]:  # reference :151-152 (bicubic, (shape[1], shape[2]))
            return fast_image_reshape(self.face_region_only_mask, like.shape[1], like.shape[2])
        return self.face_region_only_mask

    def pairwise_texture_loss(self, tx1, tx2):
        """tx1, tx2 [3,T,T] (already multiplied by the common visibility mask, as at the reference's call site)."""
        return _TexPairLossFn.apply(tx1, tx2, None, None, self._face_mask(tx1)[0])

    def texture_pairs_loss(self, textures, tx_masks, random_pairs=None):
        """The loop of tex_sp_intrp_loss (:166-176) on textures [N,3,T,T] and visibility masks [N,1,T,T]: the masking
        `textures[i] * (mask_i * mask_j)` is folded into the kernel.  random_pairs=None draws them like the reference
        (np.random.choice(len(pairs), max_num, replace=False))."""
        if random_pairs is None:
            random_pairs = self.pairs[np.random.choice(len(self.pairs), self.max_num, replace=False)]
        f = self._face_mask(textures[0])[0]
        loss = 0
        for i, j in random_pairs:
            loss = loss + _TexPairLossFn.apply(textures[i], textures[j], tx_masks[i], tx_masks[j], f)
        return 16 * loss / len(random_pairs)

    def tex_sp_intrp_loss(self, flame_batch, generator, step, alpha, max_ids, normal_maps_as_cond=True,
                          use_posed_constant_input=False, rendered_flame_as_condition=True):
        if self.flm_tex_dec is None or self.render_condition is None:
            raise ops._lib.GifHipError("tex_sp_intrp_loss needs a FLAME texture decoder and a condition renderer (FLAME assets "
                                       "are not part of this repository): pass flm_tex_dec= and render_condition=, or call "
                                       "texture_pairs_loss(textures, tx_masks) directly")
        flame_batch = flame_batch[:self.max_num]
        rend_flm, norma_map_img = self.render_condition(flame_batch)
        gen_in = torch.cat((rend_flm, norma_map_img), dim=1)
        fixed_identities = torch.ones(flame_batch.shape[0], dtype=torch.long, device=gen_in.device) * np.random.randint(0, max_ids)
        generated_image = generator(gen_in, pose=None, step=step, alpha=alpha, input_indices=fixed_identities)[-1]
        textures, tx_masks = self.flm_tex_dec(generated_image, flame_batch)
        return self.texture_pairs_loss(textures, tx_masks)
