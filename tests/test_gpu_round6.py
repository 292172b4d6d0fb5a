"""-m gpu, round 6: the texture-space interpolation loss INSIDE the training step (train.py:222-238,
loss_functions/losses.py:162-243) against the CPU oracle (oracle/train_ref.py + texture_ref.py + texture_loss_ref.py)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import assert_grads_close
from test_gpu_models import _build_d, _build_g

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _face_mask(n):
    ys, xs = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    return torch.from_numpy((((xs / 0.8) ** 2 + (ys / 0.9) ** 2) <= 1).astype(np.float32))[None, None]


@pytest.mark.parametrize("adaptive,mask_size", [(False, 256), (True, 128)])
def test_g_step_texture_interpolation_loss_vs_oracle(adaptive, mask_size):
    """One full training iteration with GifTrainer(texture_loss=...) and FLAME labels: the generator loss (adversarial +
    16 * mean pair loss, optionally rescaled adaptively, train.py:236-237) and EVERY generator parameter gradient against the
    oracle trainer on the same rendered inputs.  The FLAME layer is the synthetic stand-in (gif_amd.data.SyntheticFlame); mesh,
    normals and the rendered condition are produced by the HIP path (each oracle-checked on its own: test_gpu_kernels.py) and
    handed to the oracle as data.  mask_size 128 exercises the bicubic face-mask resize of losses.py:151-152."""
    from oracle import stylegan2_ref as R
    from oracle import texture_loss_ref as TL
    from oracle.train_ref import RefTrainer
    from gif_amd import data, losses, render
    from gif_amd.texture_space import FlameTextureSpace
    from gif_amd.train_step import GifTrainer

    dev = torch.device("cuda")
    m = np.load(os.path.join(ROOT, "tests", "golden", "body_mesh.npz"))
    B, res, step, vocab = 8, 32, 3, 16  # (minibatch-stddev groups of 4: the batch must be a multiple)
    torch.manual_seed(0)
    G, G_ema, D = _build_g(vocab=vocab), _build_g(vocab=vocab), _build_d(res)
    g_sd = R.seeded_state_dict(G.state_dict(), 71)
    d_sd = R.seeded_state_dict(D.state_dict(), 72)
    for mod, sd in ((G, g_sd), (G_ema, g_sd), (D, d_sd)):
        mod.load_state_dict(sd, strict=True)

    flame = data.SyntheticFlame(m["vertices"], dev, seed=3)
    faces = torch.from_numpy(m["faces"]).to(dev)
    td = data.synthetic_texture_data(m["faces"], fill=0.3, seed=4)
    vtx_tex = torch.rand(m["vertices"].shape[0], 3, generator=torch.Generator().manual_seed(5)).to(dev)
    renderer = render.FlameConditionRenderer(flame, faces, vtx_tex, res, res)
    tex_dec = FlameTextureSpace(td, None, flame=flame, faces=faces).to(dev)
    face_mask = _face_mask(mask_size)
    tl = losses.InterpolatedTextureLoss(B, face_mask, flm_tex_dec=tex_dec, render_condition=renderer)
    tr = GifTrainer(G.cuda(), D.cuda(), G_ema.cuda(), step=step, alpha=1.0, r1_every=0, texture_loss=tl,
                    adaptive_interp_loss=adaptive, max_ids=vocab, lr=0.0)
    ref = RefTrainer(g_sd, d_sd, res_step=step, size=res, r1_every=0, lr=0.0)  # lr 0: D is the same network on both sides in the G step

    gen = torch.Generator().manual_seed(73)
    real = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
    cond = torch.rand(B, 6, res, res, generator=gen) * 2 - 1
    idx = torch.randint(0, vocab, (B,), generator=gen)
    flm = data.synthetic_flame_labels(B, dev, torch.Generator(device=dev).manual_seed(74))
    # the camera of the synthetic labels is made for the reference's head mesh: keep the body mesh inside the image
    flm[:, 156] = 0.9
    flm[:, 157:159] = 0.0

    np.random.seed(11)
    d_got, g_got = tr.step(0, real.cuda(), cond.cuda(), idx.cuda(), flame_batch=flm)

    # the oracle replays the three np.random draws of the product path, in its order: train.py:225 (uniform), losses.py:224
    # (randint), losses.py:166 (choice)
    np.random.seed(11)
    t = np.random.uniform(0, 1)
    ident = np.random.randint(0, vocab)
    pairs = TL.all_pairs(B)[np.random.choice(len(TL.all_pairs(B)), B - 1, replace=False)]
    flm_i = losses.interpolate_flame_labels(flm, t)
    with torch.no_grad():
        verts, v_ndc, cam = renderer.vertices(flm_i)
        normals = render.vertex_normals(v_ndc, faces)
        gen_in = torch.cat(renderer(flm_i), dim=1)
    tex = {"gen_in": gen_in.cpu(), "identities": torch.full((B - 1,), ident, dtype=torch.long), "verts": verts.cpu(),
           "normals": normals.cpu(), "cam": cam.cpu(), "texture_data": td, "face_mask": face_mask, "pairs": pairs}
    d_ref, g_ref = ref.step(0, real, cond, idx, tex=tex, adaptive_interp_loss=adaptive)
    interp_ref = ref.texture_interp_loss(tex).item()
    assert interp_ref > 1e-3, interp_ref  # the fixture must make the term matter (visible texels in common)

    assert abs(d_got.item() - d_ref.item()) < 2e-3 * max(1.0, abs(d_ref.item())), (d_got.item(), d_ref.item())
    assert abs(g_got.item() - g_ref.item()) < 2e-3 * max(1.0, abs(g_ref.item())), (g_got.item(), g_ref.item(), interp_ref)
    # gradients of the G step (the bucket keeps them until the next zero())
    names, got, want = [], [], []
    named = dict(G.named_parameters())
    for k, v in ref.g.items():
        if not v.requires_grad:
            continue
        names.append(k)
        got.append(named[k].grad)
        want.append(v.grad)
    assert_grads_close(got, want, names, tight=1e-3, loose=2e-2, what=f"G grads with texture-interp loss (adaptive={adaptive})")


_FIR_CHILD = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from gif_amd import ops
k = torch.tensor([1.0, 3.0, 3.0, 1.0], device="cuda")
k = (k[:, None] * k[None, :] / 16.0).contiguous()
h = hashlib.sha1()
for dt in (torch.float32, torch.float16):
    for B, C, H, pad0, out in [(3, 8, 37, 2, 74), (3, 8, 37, 1, 73), (2, 16, 16, 2, 32), (2, 128, 33, 2, 67), (1, 8, 5, 3, 12), (4, 24, 64, 2, 128)]:
        torch.manual_seed(H)
        x = torch.randn(B, C, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        res = torch.randn(B, C, out, out, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(C, device="cuda")
        for flip in (True, False):
            h.update(ops.upfirdn2d(x, k, 2, 1, pad0, (out, out), flip).cpu().numpy().tobytes())
        h.update(ops.upfirdn2d(x, k, 2, 1, pad0, (out, out), True, bias=bias, residual=res, act=True).cpu().numpy().tobytes())
print(h.hexdigest())
"""


@pytest.mark.gpu
def test_fir_up2_block_kernel_gives_the_bits_of_the_per_pixel_kernel():
    """fir4x4_up2_block_kernel (four output pixels of one 2 x 2 input block per lane) sums every output's products in the order of
    fir4x4_resample_kernel<2, 1>: same bits, plain and with residual + bias + leaky ReLU, odd sizes, both pad parities, fp32 and f16.
    GIF_FIR_BLOCK is read once per process: one child per kernel."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _FIR_CHILD % root], env=dict(os.environ, GIF_EXPERIMENTAL="1", GIF_FIR_BLOCK=v), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(r.stdout.strip().splitlines()[-1])
    assert out[0] == out[1] and len(out[0]) == 40
