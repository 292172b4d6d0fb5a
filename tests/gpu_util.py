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


def assert_grads_close(got, ref, names, tight, loose=5e-2, max_outlier_frac=0.10, l2_tol=None, what="grads", max_outliers=None):
    """Whole-model gradients against the oracle's autograd.

    fp32 rounding leaves a handful of near-zero pre-activations on the other side of a (leaky) ReLU than in the oracle's own
    forward pass (measured on the CPU restatement of the ops, i.e. independent of the kernels: 1 element of 131 072 in one
    layer, |pre-activation| = 8e-7).  The backward mask of that element differs, which moves the gradients of the tensors fed
    by that layer by up to a few percent of their max — it says nothing about kernels or wiring, and every per-op test
    compares exactly-linear maps at 2e-5.  So: EVERY tensor within `loose`; all but `max_outlier_frac` of the tensors within
    `tight` (at most max(2, 10 %) of them: the 256x256 whole-model test shows 3 of D's 38 and 6 of G's 223 tensors between
    3e-4 and 7e-4 — round 3 allowed 25 % and up to `loose` = 5e-2, under which a regression from a few marginal outliers to a
    fifth of the model would have passed; that test now also passes loose=2e-3, three times its observed worst tensor); and
    the relative L2 error over all parameters together within `l2_tol` (default 10 * tight).  A wrong operand,
    scale or missing term shows up as O(1) errors in whole groups of tensors and fails all three.
    `max_outliers` (round 5, review item 7): an ABSOLUTE cap on the tensors above `tight` instead of the fraction — the 256x256
    test passes its observed counts + 2 (5 of 38, 8 of 223)."""
    import torch
    errs, num, den = [], 0.0, 0.0
    for k, a, b in zip(names, got, ref):
        if b is None:
            assert a is None or a.abs().max().item() == 0, f"{what}: {k} must have no gradient"
            continue
        assert a is not None, f"{what}: {k} has no gradient (the oracle's is non-zero: max {b.abs().max().item():.3e})"
        errs.append((rel_err(a, b), k))
        d = (a.detach().float().cpu() - b.detach().float().cpu())
        num += d.pow(2).sum().item()
        den += b.detach().float().pow(2).sum().item()
    errs.sort(reverse=True)
    worst = ", ".join(f"{k} {e:.2e}" for e, k in errs[:3])
    assert errs[0][0] <= loose, f"{what}: worst tensors {worst}"
    n_out = sum(1 for e, _ in errs if e > tight)
    limit = max_outliers if max_outliers is not None else (0 if max_outlier_frac == 0 else max(2, max_outlier_frac * len(errs)))
    assert n_out <= limit, f"{what}: {n_out} of {len(errs)} tensors above {tight:.0e}: {worst}"
    l2 = (num / max(den, 1e-300)) ** 0.5
    l2_tol = 10 * tight if l2_tol is None else l2_tol
    assert l2 <= l2_tol, f"{what}: relative L2 error over all parameters {l2:.2e} > {l2_tol:.0e}"
    return errs[0][0], n_out, l2
