"""One GIF training iteration on MI355X — the loop body of /root/reference/train.py:82-250 with
nn.DataParallel (train.py:344,356,358) replaced by ONE PROCESS PER GPU and a single RCCL all-reduce of the flat
gradient bucket per optimiser step (torch.distributed backend "nccl" == RCCL over xGMI).

Semantics kept from the reference: non-saturating logistic losses, R1 (weight 5.0) on the real images every
16th iteration, Adam(lr=0.002*r, betas=(0, 0.99**r)) with r = 16/17 (D) and 4/5 (G), EMA generator with decay
0.5**(32/10000) over named parameters, G frozen during the D step and vice versa.
Semantics that change with DataParallel -> per-rank replicas: parameters/buffers are never
re-broadcast after init; minibatch-stddev groups are formed inside the per-rank batch.
"""
import math

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import losses


def requires_grad(model, flag=True):  # my_utils/generic_utils.py:58-60
    for p in model.parameters():
        p.requires_grad = flag


@torch.no_grad()
def accumulate(model1, model2, decay=0.999):  # my_utils/generic_utils.py:63-76 (EMA over named parameters)
    p1 = [p for _, p in model1.named_parameters()]
    p2 = [p for _, p in model2.named_parameters()]
    torch._foreach_mul_(p1, decay)
    torch._foreach_add_(p1, p2, alpha=1 - decay)


class FlatGradBucket:
    """All parameter gradients of a model live in ONE contiguous fp32 buffer (p.grad are views into it), so the
    data-parallel exchange is exactly one all-reduce(sum)/world per optimiser step — D: 28.86 M floats = 115.5 MB,
    G: 31.63 M = 126.5 MB — instead of DataParallel's per-forward broadcast + per-backward reduce.
    Parameters that receive no gradient in a step (e.g. G blocks above `step`) simply stay zero in the bucket."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params]
        self.group = process_group
        align = 64  # floats: every view starts on a 256-byte boundary (vectorised optimiser / RCCL paths)
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, offs):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        if dist.is_available() and dist.is_initialized():
            ws = dist.get_world_size(self.group)
            if ws > 1:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.div_(ws)


class GifTrainer:
    """Holds G, D, the EMA generator, both Adam optimisers and the two gradient buckets; step() is one iteration."""

    def __init__(self, generator, discriminator, g_running, step=6, alpha=1.0, r1_every=16, gen_reg_type='None',
                 embedding_reg_weight=0.0, lr=0.002, fused_adam=None, process_group=None,
                 reuse_generator_forward=False):
        self.G, self.D, self.G_ema = generator, discriminator, g_running
        self.res_step, self.alpha, self.r1_every = step, alpha, r1_every
        self.gen_reg_type = gen_reg_type.upper()
        self.embedding_reg_weight = embedding_reg_weight
        g_ratio, d_ratio = 4 / (4 + 1), 16 / (16 + 1)  # train.py:364-381
        on_gpu = next(generator.parameters()).is_cuda
        fused = on_gpu if fused_adam is None else fused_adam
        self.g_bucket = FlatGradBucket(generator.parameters(), process_group)
        self.d_bucket = FlatGradBucket(discriminator.parameters(), process_group)
        self.g_optim = torch.optim.Adam(generator.parameters(), lr=lr * g_ratio, betas=(0.0, 0.99 ** g_ratio), fused=fused)
        self.d_optim = torch.optim.Adam(discriminator.parameters(), lr=lr * d_ratio, betas=(0.0, 0.99 ** d_ratio), fused=fused)
        self.pl_reg = losses.PathLengthRegularizor() if self.gen_reg_type == 'PATH_LEN_REG' else None
        # Optional (off by default, NOT used by bench.py): the reference runs the generator twice per iteration on
        # identical inputs and identical weights (train.py:157 and :197 — G only changes at :243).  With this flag the
        # forward is executed once with autograd enabled; the D step consumes fake.detach(), the G step back-propagates
        # through the same graph.  Bit-identical losses and updates, one generator forward (9 % of the FLOPs) less.
        self.reuse_generator_forward = reuse_generator_forward
        self.g_running_decay = 0.5 ** (32 / (10 * 1000))
        self.G_ema.train(False)
        requires_grad(self.G, False)  # train.py:68
        requires_grad(self.D, True)

    def d_step(self, i, real_image, cond, input_indices, fake=None):
        """train.py:82-178"""
        G, D = self.G, self.D
        requires_grad(D, True)
        self.d_bucket.zero()
        r1_step = bool(self.r1_every) and (i + 1) % self.r1_every == 0
        # the reference marks the real image as requiring grad on every iteration (train.py:135-136) but only uses the
        # gradient on R1 iterations; requesting it only then skips a dead dgrad of D's first layer otherwise
        real_image = real_image.detach().requires_grad_(r1_step)
        real_scores, _ = D([real_image], condition=cond, step=self.res_step, alpha=self.alpha)
        real_loss = F.softplus(-real_scores).mean()
        if r1_step:
            real_loss = real_loss + losses.grad_penalty_loss([real_image], real_scores, step=None).mean()
        if fake is None:
            with torch.no_grad():  # the reference detaches the fake image right after the forward (train.py:160)
                fake = G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)[0]
        fake_scores, _ = D([fake.detach()], condition=cond, step=self.res_step, alpha=self.alpha)
        fake_loss = F.softplus(fake_scores).mean()
        (real_loss + fake_loss).backward()
        self.d_bucket.all_reduce_mean()
        self.d_optim.step()
        return (real_loss + fake_loss).detach()

    def g_step(self, cond, input_indices, fake=None):
        """train.py:189-252"""
        G, D = self.G, self.D
        requires_grad(G, True)
        requires_grad(D, False)
        self.g_bucket.zero()
        if fake is None:
            fake = G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)
        pred, _ = D(fake, condition=cond.detach(), step=self.res_step, alpha=self.alpha)
        loss = F.softplus(-pred).mean()
        if self.pl_reg is not None:
            loss = loss + 2 * self.pl_reg.path_length_reg(G, step=self.res_step, alpha=self.alpha,
                                                          input_indices=input_indices, cond=cond)
        if self.embedding_reg_weight:
            loss = loss + self.embedding_reg_weight * losses.l2_reg(G.z_to_w)
        loss.backward()
        self.g_bucket.all_reduce_mean()
        self.g_optim.step()
        accumulate(self.G_ema, G, self.g_running_decay)
        requires_grad(G, False)
        return loss.detach()

    def step(self, i, real_image, cond, input_indices):
        if self.reuse_generator_forward:
            requires_grad(self.G, True)
            fake = self.G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)
            d_loss = self.d_step(i, real_image, cond, input_indices, fake=fake[0])
            g_loss = self.g_step(cond, input_indices, fake=fake)
            return d_loss, g_loss
        d_loss = self.d_step(i, real_image, cond, input_indices)
        g_loss = self.g_step(cond, input_indices)
        return d_loss, g_loss


# Contraction-only MACs per image at 256x256 measured on the reference: used for the roofline line
F_G_256, F_D_256 = 52.33e9, 46.58e9


def flops_per_image(res=256, r1_every=16):
    table = {64: (17.34e9, 16.46e9), 128: (33.76e9, 31.51e9), 256: (F_G_256, F_D_256), 512: (75.82e9, 61.69e9),
             1024: (111.71e9, 76.87e9)}
    fg, fd = table[res]
    fl = 2 * (4 * fg + 8 * fd)
    if r1_every:
        fl += 2 * 3 * fd / r1_every
    return fl
