"""Regularisers of the GIF training step on the HIP path — mirror of loss_functions/losses.py:87-124.

grad_penalty_loss (R1) keeps the reference signature; the per-sample squared norm is a HIP reduction
(gif_sqnorm_per_sample_f32) wrapped so that it stays differentiable (the penalty is back-propagated through D's
backward: every D layer is an any-order autograd Function, gif_amd/functional.py).
"""
import numpy as np
import torch
from torch.autograd import Function, grad

from . import ops


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


def grad_penalty_loss(inputs, outs, step):
    """R1: sum over inputs of w * ||d sum(outs) / d input||^2 per sample (losses.py:87-99).  Returns [B]."""
    grad_penalty = 0
    for inp_idx, inpt in enumerate(inputs):
        grad_real = grad(outputs=outs.sum(), inputs=inpt, create_graph=True)[0]
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
            pl_grads = grad(outputs=torch.sum(fake * noise), inputs=style)[0]
            pl_lengths = torch.mean(torch.sqrt(sqnorm_per_sample(pl_grads)))
            self.pl_moving_mean = self.pl_moving_mean + self.pl_decay * pl_lengths - self.pl_moving_mean
            return torch.pow(pl_lengths - self.pl_moving_mean, 2)
        noise = torch.randn(fake.shape, device=dev) / np.sqrt(fake.shape[2] * fake.shape[3])
        pl_grads = grad(outputs=torch.sum(fake * noise), inputs=style, create_graph=True)[0]
        pl_lengths = torch.sqrt(sqnorm_per_sample(pl_grads))
        mean = self.pl_moving_mean + self.pl_decay * (pl_lengths.mean().detach() - self.pl_moving_mean)
        self.pl_moving_mean = mean
        return (pl_lengths - mean).pow(2).mean()
