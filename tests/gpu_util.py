"""Helpers for the -m gpu parity tests (HIP path vs the CPU oracle)."""
import torch
import torch.nn.functional as F


def pad4(c):
    return (c + 3) // 4 * 4


def dev(x, requires_grad=False):
    """CPU NCHW tensor -> device, channels zero-padded to a multiple of 4, NHWC memory."""
    c = x.shape[1]
    if pad4(c) != c:
        x = F.pad(x, (0, 0, 0, 0, 0, pad4(c) - c))
    y = x.detach().cuda().contiguous(memory_format=torch.channels_last)
    return y.requires_grad_(requires_grad)


def host(y, c=None):
    y = y.detach()
    if c is not None:
        y = y[:, :c]
    return y.cpu().contiguous()


def rel_err(got, ref):
    ref = ref.detach().float().cpu()
    got = got.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def assert_close(got, ref, tol, what=""):
    e = rel_err(got, ref)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e} (ref max {ref.abs().max().item():.3e})"
