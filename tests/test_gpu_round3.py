"""-m gpu, round 3: the re-distributed rasteriser (lane / wave / queued-tile face classes), the gradient-producer epilogue
fusions (leaky-ReLU backward, bias gradient and modulation gradient inside the kernel that produces the gradient), and the
adversarial accuracy cases of the bf16x3 contraction mode."""
import os
import sys

import numpy as np
import pytest
import torch

from gpu_util import dev  # noqa: F401

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from gif_amd import _lib
    _lib.load()


# ------------------------------------------------------------------------------------------------ rasteriser work classes
def _mixed_mesh(batch, seed):
    """body.obj + screen-filling back-drop faces (queued class) + random medium faces (wave class) per image."""
    from tools.raster_bench import body_mesh, random_medium, with_backdrop
    v, f = with_backdrop(*body_mesh(batch, seed), n_big=4)
    mv, mf = random_medium(batch, nfaces=400, size=0.2, seed=seed + 1)
    mv[..., 2] = mv[..., 2] * 0.5 + v[..., 2].mean()  # interleave in depth with the body
    return np.concatenate([v, mv], 1), np.concatenate([f, mf + v.shape[1]], 1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rasterize_all_face_classes_bit_exact(dtype):
    from gif_amd import standard_rasterize as sr
    from oracle import rasterize_oracle as ro
    B, H, W = 3, 256, 256
    v, f = _mixed_mesh(B, 5)
    fv = ro.face_vertices(ro.to_image_space(v, H, W), f).astype(dtype)
    bb = (np.floor(fv[..., 0].max(-1)).clip(0, W - 1) - np.ceil(fv[..., 0].min(-1)).clip(0, W - 1) + 1).clip(0) * \
         (np.floor(fv[..., 1].max(-1)).clip(0, H - 1) - np.ceil(fv[..., 1].min(-1)).clip(0, H - 1) + 1).clip(0)
    assert (bb <= 16).sum() > 1000 and ((bb > 16) & (bb <= 4096)).sum() > 300 and (bb > 4096).sum() >= 4 * B, "all three classes present"
    d0 = np.zeros((B, H, W), dtype) + dtype(1e6)
    t0 = np.zeros((B, H, W), np.int32) - 1
    b0 = np.zeros((B, H, W, 3), dtype)
    ro.standard_rasterize(fv, d0, t0, b0, H, W)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    d1 = torch.zeros(B, H, W, device="cuda", dtype=tdt) + 1e6
    t1 = torch.zeros(B, H, W, device="cuda", dtype=torch.int32) - 1
    b1 = torch.zeros(B, H, W, 3, device="cuda", dtype=tdt)
    fvt = torch.from_numpy(fv).cuda()
    it = np.int64 if dtype == np.float64 else np.int32
    for _ in range(2):  # second call on the filled buffers: idempotent
        sr.standard_rasterize(fvt, d1, t1, b1, H, W)
        assert np.array_equal(t1.cpu().numpy(), t0), "face indices"
        assert np.array_equal(d1.cpu().numpy().view(it), d0.view(it)), "depth bits"
        assert np.array_equal(b1.cpu().numpy().view(it), b0.view(it)), "barycentric bits"
    assert (t0 >= 0).mean() > 0.9  # the back-drop covers the image


def test_rasterize_big_face_queue_overflow():
    """More queued-class faces than the queue holds (capacity max(1024, B*H*W/64)): the surplus is walked by its wave instead —
    same pixels, same bits."""
    from gif_amd import standard_rasterize as sr
    from oracle import rasterize_oracle as ro
    B, H, W, F = 1, 128, 128, 1500
    rng = np.random.RandomState(3)
    c = rng.uniform(40, 88, (B, F, 1, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (B, F, 1)).astype(np.float32) + np.array([0, 2.1, 4.2], np.float32)
    xy = c + 60 * np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
    z = rng.uniform(1, 3, (B, F, 3, 1)).astype(np.float32)
    fv = np.ascontiguousarray(np.concatenate([xy, z], -1))
    d0, t0, b0 = ro.new_buffers(B, H, W)
    ro.standard_rasterize(fv, d0, t0, b0, H, W)
    assert (t0 >= 0).mean() > 0.5
    d1, t1, b1 = sr.new_buffers(B, H, W, "cuda")
    sr.standard_rasterize(torch.from_numpy(fv).cuda(), d1, t1, b1, H, W)
    assert np.array_equal(t1.cpu().numpy(), t0)
    assert np.array_equal(d1.cpu().numpy().view(np.int32), d0.view(np.int32))
    assert np.array_equal(b1.cpu().numpy().view(np.int32), b0.view(np.int32))
