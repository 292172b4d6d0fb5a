"""On-device input pipeline pieces — SURVEY §8(f) row 3: the reference feeds its training loop from an LMDB-backed
torch DataLoader (dataset_loaders.py:390-397) and resizes images with fast_image_reshape (dataset_loaders.py:26-34).
Here batches are produced and resized on the GPU so that a multi-GPU run is never host-bound.

* fast_image_reshape — same name, arguments and semantics as the reference (incl. its (width_out, height_out) argument
  order quirk and the optional clamp to the input range), on the HIP resize kernel; differentiable.
* SyntheticBatches  — FFHQ-shaped synthetic (real image, 6-channel rendered condition, embedding index) batches drawn on
  device from a per-rank generator: what bench.py and the parity tests feed the train step with (no dataset on the box).
The LMDB wire format itself ('{res}-{idx:05d}' keys, prepare_lmdb/create_deca_rendered_lmdb.py:78-93) is out of scope:
lmdb is not installed and no dataset is available.
"""
import torch
from torch.autograd import Function

from . import ops


class _ResizeFn(Function):
    @staticmethod
    def forward(ctx, x, out_hw, mode):
        ctx.in_hw, ctx.out_hw, ctx.mode = tuple(x.shape[2:]), tuple(out_hw), mode
        return ops.resize(x, out_hw, mode)

    @staticmethod
    def backward(ctx, gy):
        return _ResizeBwdFn.apply(gy, ctx.in_hw, ctx.out_hw, ctx.mode), None, None


class _ResizeBwdFn(Function):  # linear map: its backward is the forward resize again
    @staticmethod
    def forward(ctx, gy, in_hw, out_hw, mode):
        ctx.out_hw, ctx.mode = out_hw, mode
        return ops.resize(gy, None, mode, backward_to=in_hw)

    @staticmethod
    def backward(ctx, ggx):
        return _ResizeFn.apply(ggx, ctx.out_hw, ctx.mode), None, None, None


def resize_image(x, out_hw, mode='bilinear'):
    """F.interpolate(x, size=out_hw, mode=mode, align_corners=False) for any ratio, up or down (fp32 [B,C,H,W]; NCHW result);
    differentiable to any order."""
    return _ResizeFn.apply(x, tuple(out_hw), mode)


def fast_image_reshape(in_img_batch, height_out, width_out, non_diff_allowed=False, mode='bicubic'):
    """Reference dataset_loaders.py:26-34.  Like the reference, the target size is passed to interpolate as
    (width_out, height_out) — i.e. rows = width_out — which only matters for non-square targets."""
    resize_img = _ResizeFn.apply(in_img_batch, (width_out, height_out), mode)
    if non_diff_allowed:
        min_pix = in_img_batch.min().item()
        max_pix = in_img_batch.max().item()
        resize_img = resize_img.clamp(min=min_pix, max=max_pix)
    return resize_img


class SyntheticBatches:
    """Endless iterator of device-resident synthetic training batches with the shapes and value ranges of the reference's
    FFHQ + DECA-render dataset items (dataset_loaders.py:300-378): real image in [-1,1] [B,3,R,R], rendered condition
    (texture render + normal map) in [-1,1] [B,6,R,R], dataset index int64 [B] < vocab."""

    def __init__(self, batch_size, resolution, vocab, device, seed=1234, rank=0):
        self.B, self.R, self.vocab, self.device = batch_size, resolution, vocab, device
        self.gen = torch.Generator(device=device).manual_seed(seed + rank)  # every rank draws its own data

    def __iter__(self):
        return self

    def __next__(self):
        real = torch.rand(self.B, 3, self.R, self.R, device=self.device, generator=self.gen) * 2 - 1
        cond = torch.rand(self.B, 6, self.R, self.R, device=self.device, generator=self.gen) * 2 - 1
        idx = torch.randint(0, self.vocab, (self.B,), device=self.device, generator=self.gen)
        return real, cond, idx
