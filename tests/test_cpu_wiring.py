"""CPU tier: the autograd WIRING of gif_amd (functional.py Functions, layers, generator, discriminator, losses, GifTrainer)
against the oracle, with every HIP launch replaced by its ATen contract (tests/cpu_ops.py).  Catches, without a GPU, what a
kernel test cannot: a backward that calls the wrong op / operand / scale, a gradient that silently loses its history under
create_graph (round-1 advisor finding), the data-parallel trainer's synchronisation.  The kernels behind the same entry points
are checked against the same oracle in the -m gpu tier."""
import contextlib
import io
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cpu_ops
from gpu_util import assert_close, assert_grads_close
from oracle import stylegan2_ref as R


@pytest.fixture
def cpu(monkeypatch):
    cpu_ops.install(monkeypatch)


def _build_g(vocab=16):
    from gif_amd.generator import StyledGenerator
    with contextlib.redirect_stdout(io.StringIO()):
        return StyledGenerator(embedding_vocab_size=vocab, rendered_flame_ascondition=True, normal_maps_as_cond=True)


def _build_d(size):
    from gif_amd.discriminator import Discriminator
    return Discriminator(size=size, num_color_chnls=9)


def _leaves(sd):
    return {k: (v.clone().requires_grad_(True) if (not k.endswith('kernel') and 'embd_weight' not in k) else v.clone())
            for k, v in sd.items()}


def _seeded(model, seed):
    sd = R.seeded_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd, strict=True)
    for p in model.parameters():
        p.requires_grad_(True)
    return sd


def test_generator_and_discriminator_first_order_vs_oracle(cpu):
    torch.manual_seed(0)
    g, d = _build_g(), _build_d(16)
    gsd, dsd = _seeded(g, 1), _seeded(d, 101)
    gen = torch.Generator().manual_seed(201)  # a draw without activation sign flips (see gpu_util.assert_grads_close)
    cond = torch.rand(4, 6, 16, 16, generator=gen) * 2 - 1
    idx = torch.tensor([1, 5, 9, 13])
    gl, dl = _leaves(gsd), _leaves(dsd)
    fake_r = R.generator_forward(gl, cond, 2, idx)
    loss_r = F.softplus(-R.discriminator_forward(dl, fake_r, cond, 16)).mean()
    fake = g(cond, None, step=2, alpha=1, input_indices=idx)
    assert_close(fake[0], fake_r, 1e-5, "G forward")
    loss = F.softplus(-d(fake, condition=cond)[0]).mean()
    assert abs(loss.item() - loss_r.item()) < 1e-5
    gk = [k for k, v in gl.items() if v.requires_grad]
    ref = torch.autograd.grad(loss_r, [gl[k] for k in gk], allow_unused=True)
    named = dict(g.named_parameters())
    got = torch.autograd.grad(loss, [named[k] for k in gk], allow_unused=True)
    assert_grads_close(got, ref, gk, tight=1e-4, max_outlier_frac=0.0, what="G grads through D")


def test_gradient_epilogue_fusions_equal_the_standalone_passes(cpu, monkeypatch):
    """The activation ports (leaky-ReLU backward + bias gradient + modulation gradient delivered by the kernel that produces
    the gradient, functional.ActPort / ops.GradFuse) against the same model with GIF_FUSE_GRAD off: same parameter gradients
    of G through D, and the fused routes were really taken (the dot-product fusion is forced on: at 16x16 the kernels' tile
    constraint would route it to the stand-alone pass, the contract restatement has no such constraint)."""
    from gif_amd import ops
    torch.manual_seed(0)
    g, d = _build_g(), _build_d(16)
    _seeded(g, 1), _seeded(d, 101)
    gen = torch.Generator().manual_seed(11)
    cond = torch.rand(4, 6, 16, 16, generator=gen) * 2 - 1
    idx = torch.tensor([1, 5, 9, 13])
    params = list(g.parameters()) + list(d.parameters())
    calls = {"mask": 0, "dot": 0, "colsum": 0, "bias_act_bwd": 0, "mul_reduce": 0, "colsum_op": 0}
    real_fuse = ops.GradFuse

    class CountingFuse(real_fuse):
        def __init__(self, **kw):
            super().__init__(**kw)
            calls["mask"] += self.mask_src is not None
            calls["dot"] += self.dot_src is not None
            calls["colsum"] += self.want_colsum

    def counted(name):
        fn = getattr(ops, name)

        def wrapped(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return wrapped

    monkeypatch.setattr(ops, "GradFuse", CountingFuse)
    monkeypatch.setattr(ops, "bias_act_bwd", counted("bias_act_bwd"))
    monkeypatch.setattr(ops, "mul_reduce", counted("mul_reduce"))
    plain_colsum = ops.colsum

    def colsum_counted(x):
        calls["colsum_op"] += 1
        return plain_colsum(x)
    monkeypatch.setattr(ops, "colsum", colsum_counted)

    def grads(fused):
        monkeypatch.setattr(ops, "FUSE_GRAD", fused)
        monkeypatch.setattr(ops, "dot_fusable", (lambda H, W, dtype=torch.float32: True) if fused else (lambda H, W, dtype=torch.float32: False))
        for k in calls:
            calls[k] = 0
        fake = g(cond, None, step=2, alpha=1, input_indices=idx)
        loss = F.softplus(-d(fake, condition=cond)[0]).mean()
        out = torch.autograd.grad(loss, params, allow_unused=True)
        return out, dict(calls)

    ref, c0 = grads(False)
    got, c1 = grads(True)
    assert c0["mask"] == 0 and c0["dot"] == 0 and c0["bias_act_bwd"] > 10 and c0["mul_reduce"] > 8
    assert c1["mask"] > 10 and c1["dot"] >= 8 and c1["colsum"] > 5, c1
    assert c1["bias_act_bwd"] < c0["bias_act_bwd"] // 2, (c0, c1)
    # the last condition-noise conv of every StyledConv (conv + bias, no activation) gets its bias gradient from the layer it feeds
    assert c0["colsum_op"] - c1["colsum_op"] >= 5, (c0, c1)  # five StyledConvs at step 2
    n = 0
    for a, b, (k, _) in zip(got, ref, list(g.named_parameters()) + list(d.named_parameters())):
        assert (a is None) == (b is None), k
        if b is not None:
            assert_close(a, b, 2e-5, k)
            n += 1
    assert n > 80


def test_activation_ports_do_not_leak_the_graph(cpu):
    """The port protocol hangs Python attributes on activation tensors; a reference cycle through them would keep every
    iteration's activations alive (found on the GPU as an out-of-memory after a few steps).  The number of live tensors must
    not grow from one iteration to the next — with the cyclic garbage collector off."""
    import gc
    torch.manual_seed(0)
    g, d = _build_g(), _build_d(16)
    cond = torch.rand(2, 6, 16, 16) * 2 - 1
    idx = torch.tensor([1, 5])

    def live():
        return sum(1 for o in gc.get_objects() if isinstance(o, torch.Tensor))

    gc.collect()
    gc.disable()
    try:
        counts = []
        for _ in range(3):
            fake = g(cond, None, step=2, alpha=1, input_indices=idx)
            F.softplus(-d(fake, condition=cond)[0]).mean().backward()
            del fake
            counts.append(live())
    finally:
        gc.enable()
    assert counts[1] == counts[2], counts


def test_r1_double_backward_vs_oracle(cpu):
    from gif_amd import losses
    torch.manual_seed(0)
    d = _build_d(16)
    dsd = _seeded(d, 4)
    gen = torch.Generator().manual_seed(5)
    img = torch.rand(4, 3, 16, 16, generator=gen) * 2 - 1
    cond = torch.rand(4, 6, 16, 16, generator=gen) * 2 - 1
    dl = _leaves(dsd)
    ir = img.clone().requires_grad_(True)
    sr = R.discriminator_forward(dl, ir, cond, 16)
    pen_r = R.grad_penalty_loss([ir], sr)
    keys = [k for k, v in dl.items() if v.requires_grad]
    ref = torch.autograd.grad(F.softplus(-sr).mean() + pen_r.mean(), [dl[k] for k in keys])
    ih = img.clone().requires_grad_(True)
    sh, _ = d([ih], condition=cond)
    # the inner gradient (d scores / d image, create_graph) must not compute D's weight gradients (functional.inputs_only_backward)
    from gif_amd import ops
    n_wgrad = [0]
    real_wgrad = ops.conv_wgrad

    def counted_wgrad(*a, **k):
        n_wgrad[0] += 1
        return real_wgrad(*a, **k)
    ops.conv_wgrad = counted_wgrad
    try:
        pen = losses.grad_penalty_loss([ih], sh, step=None)
        assert n_wgrad[0] == 0, "the R1 inner pass computed weight gradients"
    finally:
        ops.conv_wgrad = real_wgrad
    assert_close(pen, pen_r.detach(), 1e-4, "R1 penalty")
    named = dict(d.named_parameters())
    got = torch.autograd.grad(F.softplus(-sh).mean() + pen.mean(), [named[k] for k in keys])
    assert_grads_close(got, ref, keys, tight=1e-4, what="D grads through the R1 double backward")


def test_generator_recorded_backward_path_length_and_direct_grad(cpu, monkeypatch):
    """The generator's fused ops switch to a recorded (any-order) backward under create_graph: StyleGAN2-form path-length
    penalty and DIRECT_GRAD_REG (train.py:203-215), values AND parameter gradients vs the oracle's autograd."""
    from gif_amd import losses
    torch.manual_seed(0)
    g = _build_g()
    gsd = _seeded(g, 6)
    gen = torch.Generator().manual_seed(7)
    B = 2
    cond = torch.rand(B, 6, 16, 16, generator=gen) * 2 - 1
    style = torch.randn(B, 512, generator=gen)
    noise = torch.randn(B, 3, 16, 16, generator=gen)
    gl = _leaves(gsd)
    z = style.clone().requires_grad_(True)
    fake_r = R.generator_forward(gl, cond, 2, z)
    (pg_r,) = torch.autograd.grad((fake_r * noise / np.sqrt(16 * 16)).sum(), z, create_graph=True)
    len_r = torch.sqrt(pg_r.pow(2).sum(1))
    mean_r = 0.01 * len_r.mean().detach()
    pen_r = (len_r - mean_r).pow(2).mean()
    keys = [k for k, v in gl.items() if v.requires_grad and not any(f".{i}." in k for i in (3, 4, 5, 6, 7, 8))]
    ref = torch.autograd.grad(pen_r, [gl[k] for k in keys], allow_unused=True)
    draws = [style, noise]

    def replay(*a, **k):
        t = draws.pop(0)
        return t.clone().requires_grad_(k.get("requires_grad", False))

    monkeypatch.setattr(torch, "randn", replay)
    reg = losses.PathLengthRegularizor(reference_semantics=False)
    pen = reg.path_length_reg(g, step=2, alpha=1.0, input_indices=torch.zeros(B, dtype=torch.long), cond=cond)
    monkeypatch.undo()
    cpu_ops.install(monkeypatch)
    assert pen.requires_grad
    assert abs(pen.item() - pen_r.item()) < 2e-2 * abs(pen_r.item()), (pen.item(), pen_r.item())
    named = dict(g.named_parameters())
    got = torch.autograd.grad(pen, [named[k] for k in keys], allow_unused=True)
    assert sum(1 for b in ref if b is not None and b.abs().max() > 0) > 40
    assert_grads_close(got, ref, keys, tight=2e-4, what="d PL penalty / d parameters (recorded backward of G)")
    # DIRECT_GRAD_REG
    gl = _leaves(gsd)
    idx = torch.tensor([3, 11])
    c_r = cond.clone().requires_grad_(True)
    pen_r = R.grad_penalty_loss([c_r], R.generator_forward(gl, c_r, 2, idx).pow(2)).mean()
    ref = torch.autograd.grad(pen_r, [gl[k] for k in keys], allow_unused=True)
    c_h = cond.clone().requires_grad_(True)
    fake = g(c_h, None, step=2, alpha=1, input_indices=idx)
    pen = losses.grad_penalty_loss([c_h], torch.pow(fake[-1], 2), step=None).mean()
    assert abs(pen.item() - pen_r.item()) < 2e-2 * abs(pen_r.item())
    got = torch.autograd.grad(pen, [named[k] for k in keys], allow_unused=True)
    assert_grads_close(got, ref, keys, tight=2e-4, what="d direct-grad penalty / d parameters")


def test_trainer_trajectory_vs_oracle_trainer_on_cpu(cpu):
    """GifTrainer (torch Adam on CPU) vs oracle/train_ref.py: two iterations incl. an R1 one — losses, weights, EMA."""
    from gif_amd.train_step import GifTrainer
    from oracle.train_ref import RefTrainer
    torch.manual_seed(0)
    G, G_ema, D = _build_g(), _build_g(), _build_d(16)
    g_sd = R.seeded_state_dict(G.state_dict(), 61)
    d_sd = R.seeded_state_dict(D.state_dict(), 62)
    G.load_state_dict(g_sd)
    G_ema.load_state_dict(g_sd)
    D.load_state_dict(d_sd)
    ref = RefTrainer(g_sd, d_sd, res_step=2, size=16, r1_every=2)
    tr = GifTrainer(G, D, G_ema, step=2, r1_every=2, fused_adam=False)
    gen = torch.Generator().manual_seed(63)
    for i in range(2):
        real = torch.rand(4, 3, 16, 16, generator=gen) * 2 - 1
        cond = torch.rand(4, 6, 16, 16, generator=gen) * 2 - 1
        idx = torch.randint(0, 16, (4,), generator=gen)
        d_ref, g_ref = ref.step(i, real, cond, idx)
        d_got, g_got = tr.step(i, real, cond, idx)
        assert abs(d_got.item() - d_ref.item()) < 1e-3 * max(1.0, abs(d_ref.item())), (i, d_got.item(), d_ref.item())
        assert abs(g_got.item() - g_ref.item()) < 1e-3 * max(1.0, abs(g_ref.item())), (i, g_got.item(), g_ref.item())
    # parameters above the current resolution never receive a gradient: like in the reference, Adam holds no state for them
    dead = G.generator.progression[5].st_cv1.conv.weight
    assert dead.grad is None and len(tr.g_optim.state.get(dead, {})) == 0


# ---- the REAL trainer in two data-parallel processes (gloo, CPU) -------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    try:
        import copy
        import hashlib
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        cpu_ops.install()
        from gif_amd.train_step import GifTrainer
        torch.manual_seed(1000 + rank)  # different initial weights and embedding buffers per rank
        G, G_ema, D = _build_g(), _build_g(), _build_d(16)
        G_ema.load_state_dict(G.state_dict())
        w_before = G.generator.progression[1].st_cv2.conv.weight.detach().clone()
        tr = GifTrainer(G, D, G_ema, step=2, r1_every=2, fused_adam=False)  # broadcasts rank 0's state
        w_synced = G.generator.progression[1].st_cv2.conv.weight.detach().clone()
        emb = G.image_embedding.embd_weight.detach().clone()
        gen = torch.Generator().manual_seed(9)  # the GLOBAL batch of 8, identical in both processes
        real = torch.rand(2, 8, 3, 16, 16, generator=gen) * 2 - 1
        cond = torch.rand(2, 8, 6, 16, 16, generator=gen) * 2 - 1
        idx = torch.randint(0, 16, (2, 8), generator=gen)
        sl = slice(rank * 4, rank * 4 + 4)
        halves = []
        for h in range(world):  # the two halves' D gradients WITHOUT data parallelism, on copies of the synced models
            G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
            hs = slice(h * 4, h * 4 + 4)
            rs, _ = D2([real[0, hs]], condition=cond[0, hs])
            with torch.no_grad():
                fk = G2(cond[0, hs], None, step=2, alpha=1.0, input_indices=idx[0, hs])[0]
            fs, _ = D2([fk], condition=cond[0, hs])
            gs = torch.autograd.grad(F.softplus(-rs).mean() + F.softplus(fs).mean(), list(D2.parameters()))
            halves.append(torch.cat([g.reshape(-1) for g in gs]))
        mean_halves = (halves[0] + halves[1]) / 2
        assert tr.overlap_comm
        tr.d_step(0, real[0, sl], cond[0, sl], idx[0, sl])
        tr.d_bucket.wait()
        exchanged = torch.cat([p.grad.reshape(-1) for p in D.parameters()]).clone()
        tr.g_step(cond[0, sl], idx[0, sl])
        l1 = tr.step(1, real[1, sl], cond[1, sl], idx[1, sl])  # R1 iteration
        tr.flush()
        err = ((exchanged - mean_halves).abs().max() / mean_halves.abs().max()).item()

        def digest(m):
            return hashlib.sha1(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy().tobytes()).hexdigest()

        q.put((rank, "ok", hashlib.sha1(w_before.numpy().tobytes()).hexdigest(), hashlib.sha1(w_synced.numpy().tobytes()).hexdigest(),
               hashlib.sha1(emb.numpy().tobytes()).hexdigest(), err, digest(G), digest(D), digest(G_ema), [t.item() for t in l1]))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def test_real_trainer_two_gloo_processes_on_cpu():
    """GifTrainer itself (not a stand-in model) in two gloo processes: construction-time broadcast from different per-rank
    seeds, exchanged D gradients == mean of the halves' single-process gradients, replicas bit-identical after two iterations."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] == "ok", r[2]
    (_, _, b0, s0, e0, err0, g0, d0, m0, l0), (_, _, b1, s1, e1, err1, g1, d1, m1, l1) = res
    assert b0 != b1, "ranks started from different weights"
    assert s0 == s1 == b0, "construction broadcast rank 0's parameters"
    assert e0 == e1, "construction broadcast rank 0's embedding BUFFER"
    assert err0 < 1e-5 and err1 < 1e-5, (err0, err1)
    assert g0 == g1 and d0 == d1 and m0 == m1, "replicas bit-identical after 2 iterations"
    assert all(np.isfinite(l0)) and all(np.isfinite(l1))


def test_fused_discriminator_passes_equal_the_two_calls(cpu):
    """GifTrainer.d_step runs D once on [real; fake] (minibatch-stddev per half) where train.py:142 / :169 call it twice: same
    loss, same gradients (every D parameter receives one gradient instead of two that autograd adds), R1 iterations unchanged."""
    import copy
    from gif_amd.train_step import GifTrainer
    torch.manual_seed(0)
    G, G_ema, D = _build_g(), _build_g(), _build_d(16)
    G_ema.load_state_dict(G.state_dict())
    gen = torch.Generator().manual_seed(5)
    real = torch.rand(8, 3, 16, 16, generator=gen) * 2 - 1
    cond = torch.rand(8, 6, 16, 16, generator=gen) * 2 - 1
    idx = torch.randint(0, 16, (8,), generator=gen)
    out = {}
    for fuse in (True, False):
        g, ge, d = copy.deepcopy(G), copy.deepcopy(G_ema), copy.deepcopy(D)
        tr = GifTrainer(g, d, ge, step=2, r1_every=2, fused_adam=False, fuse_d_passes=fuse)
        calls = []
        orig = d.forward
        d.forward = lambda *a, _o=orig, **k: (calls.append(a[0][0].shape[0]), _o(*a, **k))[1]
        loss0 = tr.d_step(0, real, cond, idx)
        grads = [p.grad.clone() for p in d.parameters()]
        n_plain = len(calls)
        loss1 = tr.d_step(1, real, cond, idx)  # R1 iteration: separate calls in both modes
        out[fuse] = (loss0.item(), grads, n_plain, len(calls) - n_plain, calls[:n_plain], loss1.item())
    assert out[True][2] == 1 and out[True][4] == [16], "one D call over 2 x 8 samples"
    assert out[False][2] == 2 and out[False][4] == [8, 8]
    assert out[True][3] == 2 and out[False][3] == 2, "R1 iterations keep the two calls"
    assert abs(out[True][0] - out[False][0]) < 1e-6 * max(1.0, abs(out[False][0]))
    assert abs(out[True][5] - out[False][5]) < 1e-4 * max(1.0, abs(out[False][5]))
    for a, b in zip(out[True][1], out[False][1]):
        assert_close(a, b, 2e-5, "D gradients: fused pass vs two calls")


def test_modulation_bank_is_one_call_and_equals_the_per_layer_linears(cpu, monkeypatch):
    """Generator.forward hands every ModulatedConv2d the same w (len(style) < 2): their EqualLinears run as ONE LinearBankFn call
    (reference: one call per layer, stylegan2_common_layers.py:311-313) — same image, same parameter gradients; two styles
    (mixing) fall back to the per-layer path."""
    from gif_amd import functional as GF, layers
    torch.manual_seed(0)
    g = _build_g()
    _seeded(g, 9)
    gen = torch.Generator().manual_seed(3)
    cond = torch.rand(2, 6, 16, 16, generator=gen) * 2 - 1
    idx = torch.tensor([1, 7])
    calls = []
    orig = GF.ops.linear_bank_fwd
    monkeypatch.setattr(GF.ops, "linear_bank_fwd", lambda x, ws, bs, sc: (calls.append(len(ws)), orig(x, ws, bs, sc))[1])
    out = {}
    for bank in (True, False):
        monkeypatch.setattr(layers, "_STYLE_BANK", bank)
        g.zero_grad(set_to_none=True)
        img = g(cond, None, step=2, alpha=1, input_indices=idx)[-1]
        img.pow(2).mean().backward()
        out[bank] = (img.detach().clone(), {k: p.grad.clone() for k, p in g.named_parameters() if p.grad is not None})
    assert calls == [8], calls  # 4x4: conv + ToRGB, 8x8 and 16x16: two convs + ToRGB each
    assert_close(out[True][0], out[False][0], 1e-6, "image: bank vs per-layer modulation")
    assert out[True][1].keys() == out[False][1].keys()
    for k in out[False][1]:
        assert_close(out[True][1][k], out[False][1][k], 2e-5, f"grad {k}: bank vs per-layer modulation")
    assert all(c.conv._banked is None for c in g.generator.to_rgb), "every banked s was consumed"
    # two styles: no bank
    monkeypatch.setattr(layers, "_STYLE_BANK", True)
    calls.clear()
    styles = [torch.randn(2, 512, generator=gen), torch.randn(2, 512, generator=gen)]
    noise = g._condition_pyramid(cond, 2) if hasattr(g, "_condition_pyramid") else None
    if noise is not None:
        g.generator(styles, None, noise, step=2, alpha=1)
        assert calls == []


def test_interpolate_flame_labels_and_synthetic_flame_fixtures():
    """Host logic of the texture-interpolation hook (train.py:224-227): neighbouring labels blended with one weight, light /
    texture codes of the first sample kept; the synthetic FLAME stand-in and texture-space fixture have the reference's shapes."""
    import numpy as np
    from gif_amd import data, losses
    lbl = torch.arange(4 * 236, dtype=torch.float32).view(4, 236)
    out = losses.interpolate_flame_labels(lbl, t=0.25)
    assert out.shape == (3, 236)
    assert torch.allclose(out[:, :159], lbl[:-1, :159] + 0.25 * (lbl[1:, :159] - lbl[:-1, :159]))
    assert torch.equal(out[:, 159:], lbl[:-1, 159:])
    np.random.seed(3)
    t = np.random.uniform(0, 1)
    np.random.seed(3)
    assert torch.allclose(losses.interpolate_flame_labels(lbl), losses.interpolate_flame_labels(lbl, t))
    tmpl = np.random.RandomState(0).randn(50, 3).astype(np.float32)
    flame = data.SyntheticFlame(tmpl, "cpu", seed=1)
    lab = data.synthetic_flame_labels(3, "cpu", torch.Generator().manual_seed(2))
    v, a, b = flame(shape_params=lab[:, :100], expression_params=lab[:, 100:150], pose_params=lab[:, 150:156])
    assert v.shape == (3, 50, 3) and a is None and b is None
    v0, _, _ = flame(torch.zeros(1, 100), torch.zeros(1, 50), torch.zeros(1, 6))
    assert torch.allclose(v0[0], torch.from_numpy(tmpl), atol=1e-6)  # zero parameters = the template
    R = data._rodrigues(torch.tensor([[0.0, 0.0, np.pi / 2]]))
    assert torch.allclose(R[0] @ torch.tensor([1.0, 0.0, 0.0]), torch.tensor([0.0, 1.0, 0.0]), atol=1e-6)
    faces = np.random.RandomState(1).randint(0, 50, (80, 3))
    td = data.synthetic_texture_data(faces, T=64, fill=0.5)
    n = len(td["valid_pixel_ids"])
    assert td["valid_pixel_3d_faces"].shape == (n, 3) and td["valid_pixel_b_coords"].shape == (n, 3)
    assert abs(n / 64 ** 2 - 0.5) < 0.05 and np.allclose(td["valid_pixel_b_coords"].sum(1), 1, atol=1e-5)
