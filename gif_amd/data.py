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
import numpy as np
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


def _rodrigues(r):
    """Axis-angle [B,3] -> rotation matrices [B,3,3]."""
    theta = r.norm(dim=1, keepdim=True).clamp_min(1e-8)
    k = r / theta
    K = torch.zeros(r.shape[0], 3, 3, device=r.device, dtype=r.dtype)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    s, c = torch.sin(theta)[:, :, None], torch.cos(theta)[:, :, None]
    return torch.eye(3, device=r.device, dtype=r.dtype)[None] + s * K + (1 - c) * torch.bmm(K, K)


class SyntheticFlame:
    """Stand-in for the FLAME layer of the absent `photometric_optimization` submodule (licensed assets), with the
    call signature FlameTextureSpace.forward uses (model/stg2_generator.py:362-364): (shape_params [B,100], expression_params
    [B,50], pose_params [B,6]) -> (vertices [B,V,3], None, None).  A linear blend-shape model over a template mesh (smooth,
    low-frequency displacement fields) followed by the global rotation pose_params[:, :3] (axis-angle): the same tensor shapes
    and the same amount of arithmetic as FLAME's shape / expression blend and root rotation — what a synthetic-data run of the
    texture-interpolation loss needs, NOT a face model."""

    def __init__(self, template, device, n_shape=100, n_exp=50, seed=0, amplitude=2e-3):
        g = torch.Generator().manual_seed(seed)
        t = torch.as_tensor(np.asarray(template), dtype=torch.float32)
        V = t.shape[0]

        def basis(n):
            freq = torch.randn(n, 3, generator=g) * 3.0
            phase = torch.rand(n, 1, generator=g) * 6.2831853
            direction = torch.nn.functional.normalize(torch.randn(n, 1, 3, generator=g), dim=2)
            field = torch.sin(freq @ t.t() + phase)[:, :, None] * direction  # [n,V,3]
            return (amplitude * field).reshape(n, V * 3)

        self.template = t.reshape(1, V * 3).to(device)
        self.shape_basis, self.exp_basis = basis(n_shape).to(device), basis(n_exp).to(device)

    def __call__(self, shape_params, expression_params, pose_params):
        B = shape_params.shape[0]
        v = (self.template + shape_params @ self.shape_basis + expression_params @ self.exp_basis).view(B, -1, 3)
        return torch.bmm(v, _rodrigues(pose_params[:, :3]).transpose(1, 2)), None, None


def synthetic_texture_data(faces, T=256, fill=0.6, seed=0):
    """Stand-in for cnst.flame_texture_space_dat_file (licensed): the dict FlameTextureSpace.__init__ reads
    (stg2_generator.py:348-353).  A centred disc of `fill` of the TxT texture is valid; texels are assigned to the faces in
    raster order (neighbouring texels share a face or sit on neighbouring ones, like a real UV chart) with random
    barycentric coordinates."""
    rng = np.random.RandomState(seed)
    faces = np.asarray(faces)
    ys, xs = np.meshgrid(np.arange(T), np.arange(T), indexing="ij")
    r2 = (xs - T / 2 + 0.5) ** 2 + (ys - T / 2 + 0.5) ** 2
    valid = np.flatnonzero((r2 <= fill * T * T / np.pi).reshape(-1))
    fidx = (np.arange(len(valid)) * (len(faces) / max(len(valid), 1))).astype(np.int64)
    return {"x_coords": xs.reshape(-1), "y_coords": ys.reshape(-1), "valid_pixel_ids": valid,
            "valid_pixel_3d_faces": faces[fidx], "valid_pixel_b_coords": rng.dirichlet([1, 1, 1], len(valid)).astype(np.float32)}


def synthetic_flame_labels(batch_size, device, generator=None, n_labels=159):
    """FLAME labels `flm_lbls` [B,159] in the layout of constants.INDICES (shape 0:100, expression 100:150, pose 150:156,
    camera 156:159) with magnitudes of fitted FFHQ parameters: unit-normal shape / expression codes, small rotations, an
    orthographic camera that keeps the template inside the image."""
    lbl = torch.randn(batch_size, n_labels, device=device, generator=generator)
    lbl[:, 150:156] *= 0.15
    lbl[:, 156] = 0.95 + 0.05 * lbl[:, 156].clamp(-1, 1)
    lbl[:, 157:159] = 0.02 * lbl[:, 157:159].clamp(-2, 2) + torch.tensor([0.0, 0.35], device=device)
    return lbl
