"""-m gpu: whole-model parity of the HIP generator / discriminator against (a) golden vectors produced by the real
reference and (b) the CPU oracle at the benchmark resolution.  Tolerances: north-star L_inf < 1e-3 on the generator
image (we hold 1e-4 relative), gradients 2e-4 relative to their max magnitude."""
import contextlib
import io
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_close, rel_err
from test_oracle_stylegan2 import _gold, d_state, g_state

pytestmark = pytest.mark.gpu


def _build_g(vocab=50):
    from gif_amd.generator import StyledGenerator
    with contextlib.redirect_stdout(io.StringIO()):
        return StyledGenerator(embedding_vocab_size=vocab, rendered_flame_ascondition=True, normal_maps_as_cond=True)


def _build_d(size):
    from gif_amd.discriminator import Discriminator
    return Discriminator(size=size, num_color_chnls=9)


def test_generator_golden_forward_backward():
    gold = _gold()
    c = gold["g32"]
    g = _build_g()
    g.load_state_dict(g_state(gold, c["seed"]), strict=True)  # reference-shaped state_dict, strict
    g = g.cuda()
    out = g(c["cond"].cuda(), None, step=3, alpha=1, input_indices=c["idx"].cuda())
    assert isinstance(out, list) and len(out) == 1 and out[0].shape == (2, 3, 32, 32)
    out = out[0]
    assert (out.detach().cpu() - c["out"]).abs().max().item() < 1e-3  # north-star bound
    assert_close(out, c["out"], 1e-4, "G(32x32) vs reference golden")
    loss = (out * torch.linspace(-1, 1, out.numel(), device="cuda").view_as(out)).sum()
    loss.backward()
    gen = g.generator
    checks = [(gen.const_input.input.grad, c["grad_const"]),
              (gen.progression[0].st_cv1.conv.weight.grad[0, :2], c["grad_w_4x4"]),
              (gen.progression[2].st_cv1.conv.modulation.bias.grad, c["grad_mod_b_16"]),
              (gen.progression[3].st_cv2.noise.noise_conv[4].weight.grad[:8], c["grad_noise_w_32"]),
              (gen.to_rgb[3].conv.weight.grad, c["grad_rgb_w_32"]),
              (g.z_to_w[8].bias.grad, c["grad_z_to_w_8_b"])]
    for i, (got, ref) in enumerate(checks):
        assert_close(got, ref, 4e-4, f"G grad #{i}")  # (2.6e-4 observed on the modulation-bias gradient, a sum over the batch)


def test_generator_config1_golden():
    """BASELINE config 1: step=4 (64x64), batch 4, zero condition, z fed as float32 input_indices."""
    gold = _gold()
    c = gold["g64"]
    g = _build_g()
    g.load_state_dict(g_state(gold, c["seed"]), strict=True)
    g = g.cuda().eval()
    with torch.no_grad():
        out = g(torch.zeros(4, 6, 64, 64, device="cuda"), None, step=4, alpha=1, input_indices=c["z"].cuda())[0]
    assert (out.cpu() - c["out"]).abs().max().item() < 1e-3
    assert_close(out, c["out"], 1e-4, "G config 1")


def test_discriminator_golden_scores_r1_grads():
    from gif_amd import losses
    gold = _gold()
    c = gold["d32"]
    d = _build_d(32)
    d.load_state_dict(d_state(gold, c["seed"]), strict=True)
    d = d.cuda()
    img = c["img"].cuda().requires_grad_(True)
    scores, none = d([img], condition=c["cond"].cuda())
    assert none is None and scores.shape == (4, 1)
    assert_close(scores, c["scores"], 1e-4, "D scores")
    pen = losses.grad_penalty_loss([img], scores, step=None)
    assert_close(pen, c["r1"], 2e-4, "R1 penalty")
    (F.softplus(-scores).mean() + pen.mean()).backward()
    assert_close(d.convs[0][0].weight.grad, c["grad_first_w"], 3e-4, "D grad first conv (through R1 double backward)")
    assert_close(d.convs[1].conv2[1].weight.grad[:4], c["grad_res1_conv2_w"], 3e-4, "D grad res1.conv2")
    assert_close(d.final_conv[0].weight.grad[:2], c["grad_final_conv_w"], 3e-4, "D grad final_conv")
    assert_close(d.final_linear[1].weight.grad, c["grad_lin1_w"], 3e-4, "D grad last linear")
    assert_close(img.grad, c["grad_img"], 3e-4, "grad wrt image")
    with torch.no_grad():
        s8 = d([gold["d32_b8"]["img"].cuda()], condition=gold["d32_b8"]["cond"].cuda())[0]
    assert_close(s8, gold["d32_b8"]["scores"], 1e-4, "D scores, two stddev groups")
    with pytest.raises(Exception):  # batch 6: group 4 does not divide 6 -> the reference's view() raises too
        d([torch.zeros(6, 3, 32, 32, device="cuda")], condition=torch.zeros(6, 6, 32, 32, device="cuda"))


def test_generator_and_discriminator_256_vs_oracle():
    """Benchmark resolution (step 6, 256x256), batch 2: HIP path vs the CPU oracle on identical seeded weights."""
    from oracle import stylegan2_ref as R
    torch.manual_seed(0)
    g = _build_g(vocab=16)
    sd = R.seeded_state_dict(g.state_dict(), 21)
    g.load_state_dict(sd, strict=True)
    cond = torch.rand(2, 6, 256, 256) * 2 - 1
    idx = torch.tensor([2, 9])
    with torch.no_grad():
        ref = R.generator_forward(sd, cond, 6, idx)
        got = g.cuda()(cond.cuda(), None, step=6, alpha=1, input_indices=idx.cuda())[0]
    assert (got.cpu() - ref).abs().max().item() < 1e-3, "north-star: generator output L_inf < 1e-3"
    assert_close(got, ref, 1e-4, "G(256)")
    d = _build_d(256)
    sdd = R.seeded_state_dict(d.state_dict(), 22)
    d.load_state_dict(sdd, strict=True)
    with torch.no_grad():
        sref = R.discriminator_forward(sdd, ref, cond, 256)
        sgot = d.cuda()(got, condition=cond.cuda())[0]
    assert_close(sgot, sref, 2e-4, "D(256) scores")


def test_config3_render_condition_into_train_step_with_r1_and_path_length():
    """BASELINE config 3 (reduced size for test time): mesh -> HIP vertex normals + rasteriser -> 6-channel condition ->
    one full G+D training iteration on an R1 step with the path-length regulariser enabled; checks the plumbing end to
    end (finite losses, parameters move, EMA follows)."""
    import numpy as np
    from gif_amd import render
    from gif_amd.train_step import GifTrainer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "body_mesh.npz"))
    B, R = 4, 64
    rng = np.random.RandomState(0)
    verts = []
    for i in range(B):  # small random rotations about y, as a stand-in for FLAME pose variation
        a = rng.uniform(-0.4, 0.4)
        Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        verts.append((g["vertices"] @ Rm.T).astype(np.float32))
    v = torch.from_numpy(np.stack(verts)).cuda()
    f = torch.from_numpy(g["faces"]).cuda()
    cam = torch.tensor([[0.95, 0.0, 0.35]] * B, device="cuda")
    v_ndc = render.batch_orth_proj(v, cam)
    tex = (v - v.amin(dim=1, keepdim=True)) / (v.amax(dim=1, keepdim=True) - v.amin(dim=1, keepdim=True))
    cond = render.render_condition(v_ndc, f, tex, R, R)
    assert cond.shape == (B, 6, R, R) and (cond[:, 3:].abs().sum(dim=(1, 2, 3)) > 0).all()
    torch.manual_seed(0)
    G, G_ema, D = _build_g(16).cuda(), _build_g(16).cuda(), _build_d(R).cuda()
    G_ema.load_state_dict(G.state_dict())
    tr = GifTrainer(G, D, G_ema, step=4, gen_reg_type='PATH_LEN_REG')
    w0 = G.generator.progression[4].st_cv2.conv.weight.detach().clone()
    e0 = G_ema.generator.progression[4].st_cv2.conv.weight.detach().clone()
    real = torch.rand(B, 3, R, R, device="cuda") * 2 - 1
    idx = torch.randint(0, 16, (B,), device="cuda")
    d_loss, g_loss = tr.step(15, real, cond, idx)  # i = 15 -> R1 iteration
    torch.cuda.synchronize()
    assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
    w1 = G.generator.progression[4].st_cv2.conv.weight
    assert (w1 - w0).abs().max() > 0, "generator parameters must move"
    e1 = G_ema.generator.progression[4].st_cv2.conv.weight
    decay = 0.5 ** (32 / 10000)
    assert torch.allclose(e1, decay * e0 + (1 - decay) * w1, atol=1e-6), "EMA update (generic_utils.accumulate)"


def test_less_travelled_api_paths():
    """Options every caller of the reference may use but the benchmark does not: down-sampling ModulatedConv2d,
    EqualConv2d with bias, ScaledLeakyReLU, Upsample/Downsample modules, w-truncation, mean_style mixing, tensor (non
    list) discriminator input without condition, float-z input."""
    from gif_amd import layers as L
    from gpu_util import dev, host
    from oracle import stylegan2_ref as R
    torch.manual_seed(5)
    x, st = torch.randn(2, 16, 8, 8), torch.randn(2, 512)
    m = L.ModulatedConv2d(16, 24, 3, 512, downsample=True).cuda()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = R.modulated_conv2d(x, sd["weight"], sd["modulation.weight"], sd["modulation.bias"], st, True, False,
                             sd["blur.kernel"], True)
    assert_close(host(m(dev(x), st.cuda())), ref, 3e-5, "down-sampling modulated conv")
    c = L.EqualConv2d(16, 20, 3, stride=1, padding=1, bias=True).cuda()
    with torch.no_grad():
        c.bias.normal_()
    assert_close(host(c(dev(x)), 20), R.equal_conv2d(x, c.weight.detach().cpu(), c.bias.detach().cpu(), 1, 1), 2e-5, "EqualConv2d+bias")
    assert_close(host(L.ScaledLeakyReLU(0.2)(dev(x))), F.leaky_relu(x, 0.2) * 2 ** 0.5, 1e-6, "ScaledLeakyReLU")
    k = [1, 3, 3, 1]
    assert_close(host(L.Upsample(k).cuda()(dev(x))), R.upfirdn2d(x, R.make_kernel(k) * 4, up=2, pad=(2, 1)), 1e-6, "Upsample")
    assert_close(host(L.Downsample(k).cuda()(dev(x))), R.upfirdn2d(x, R.make_kernel(k), down=2, pad=(1, 1)), 1e-6, "Downsample")
    # generator options
    gold = _gold()
    g = _build_g()
    sdg = g_state(gold, 11)
    g.load_state_dict(sdg, strict=True)
    g = g.cuda().eval()
    cond = torch.rand(2, 6, 32, 32) * 2 - 1
    idx = torch.tensor([4, 9])
    with torch.no_grad():
        w = R.z_to_w(sdg, sdg["image_embedding.embd_weight"][idx])
        mean_w = R.z_to_w(sdg, sdg["image_embedding.embd_weight"]).mean(0)
        base = g(cond.cuda(), None, step=3, input_indices=idx.cuda())[0]
        g.w_truncation_factor = 0.7
        trunc = g(cond.cuda(), None, step=3, input_indices=idx.cuda())[0]
        g.w_truncation_factor = 1.0
        # reference :278-281: w + (mean_w - w) * (1 - factor); emulate through the float-"indices" (z) path is not
        # possible (w is post-mapping), so check against the oracle synthesis driven by the truncated w
        sd2 = dict(sdg)
        wt = w + (mean_w - w) * (1.0 - 0.7)
        ref_t = _oracle_synthesis(R, sd2, cond, wt, 3)
        assert_close(trunc, ref_t, 1e-4, "w truncation")
        assert (trunc - base).abs().max() > 1e-3
        ms = torch.randn(512)
        mixed = g(cond.cuda(), None, step=3, input_indices=idx.cuda(), mean_style=ms.cuda(), style_weight=0.5)[0]
        assert_close(mixed, _oracle_synthesis(R, sd2, cond, ms + 0.5 * (w - ms), 3), 1e-4, "mean_style mixing")
    # discriminator: bare tensor input, no condition, 3 colour channels
    d = _build_d3(32)
    sdd = R.seeded_state_dict(d.state_dict(), 7)
    d.load_state_dict(sdd, strict=True)
    d = d.cuda()
    img = torch.rand(4, 3, 32, 32) * 2 - 1
    with torch.no_grad():
        assert_close(d(img.cuda())[0], R.discriminator_forward(sdd, img, None, 32), 2e-4, "D without condition")


def _build_d3(size):
    from gif_amd.discriminator import Discriminator
    return Discriminator(size=size, num_color_chnls=3)


def _oracle_synthesis(R, sd, cond, w, step):
    """Generator.forward driven by an explicit w (for the truncation / mean-style options of StyledGenerator.forward)."""
    B = cond.shape[0]
    out = sd['generator.const_input.input'].repeat(B, 1, 1, 1)
    rgb = None
    for i in range(step + 1):
        size = 4 * 2 ** i
        c_i = F.interpolate(cond, size=(size, size), mode='bilinear', align_corners=False)
        p = f'generator.progression.{i}.'
        out = R.styled_conv(sd, p + 'st_cv1.', out, w, c_i, upsample=(i != 0))
        if i != 0:
            out = R.styled_conv(sd, p + 'st_cv2.', out, w, c_i, upsample=False)
        rgb = R.to_rgb(sd, f'generator.to_rgb.{i}.', out, w, rgb)
    return rgb


def test_reuse_generator_forward_is_bit_identical():
    """GifTrainer(reuse_generator_forward=True) must give exactly the losses and parameters of the reference call order."""
    from gif_amd.train_step import GifTrainer
    res = {}
    for reuse in (False, True):
        torch.manual_seed(0)
        G, G_ema, D = _build_g(16).cuda(), _build_g(16).cuda(), _build_d(32).cuda()
        G_ema.load_state_dict(G.state_dict())
        tr = GifTrainer(G, D, G_ema, step=3, reuse_generator_forward=reuse)
        gen = torch.Generator(device="cuda").manual_seed(1)
        out = []
        for i in (14, 15):  # a plain and an R1 iteration
            real = torch.rand(4, 3, 32, 32, device="cuda", generator=gen) * 2 - 1
            cond = torch.rand(4, 6, 32, 32, device="cuda", generator=gen) * 2 - 1
            idx = torch.randint(0, 16, (4,), device="cuda", generator=gen)
            out.append([t.item() for t in tr.step(i, real, cond, idx)])
        res[reuse] = (out, G.generator.progression[3].st_cv2.conv.weight.detach().clone(),
                      D.convs[1].conv1[0].weight.detach().clone())
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])


def test_generator_discriminator_1024_vs_oracle():
    """BASELINE config 5 resolution (step 8, 1024x1024) in fp32, batch 1: every layer shape of the full progression
    (512..32 channels) through the HIP path vs the CPU oracle.  (The fp16-activation variant of config 5 is not built.)"""
    from oracle import stylegan2_ref as R
    torch.manual_seed(0)
    g = _build_g(vocab=4)
    sd = R.seeded_state_dict(g.state_dict(), 31)
    g.load_state_dict(sd, strict=True)
    cond = torch.rand(1, 6, 1024, 1024) * 2 - 1
    idx = torch.tensor([3])
    with torch.no_grad():
        ref = R.generator_forward(sd, cond, 8, idx)
        got = g.cuda()(cond.cuda(), None, step=8, alpha=1, input_indices=idx.cuda())[0]
    assert got.shape == (1, 3, 1024, 1024)
    assert (got.cpu() - ref).abs().max().item() < 1e-3
    assert_close(got, ref, 2e-4, "G(1024)")
    d = _build_d(1024)
    sdd = R.seeded_state_dict(d.state_dict(), 32)
    d.load_state_dict(sdd, strict=True)
    with torch.no_grad():
        sref = R.discriminator_forward(sdd, ref, cond, 1024)
        sgot = d.cuda()(got, condition=cond.cuda())[0]
    assert_close(sgot, sref, 3e-4, "D(1024) score")


def test_goldens_with_every_eligible_layer_on_winograd(monkeypatch):
    """The reference goldens again with the Winograd tile threshold at 0, so that EVERY stride-1 3x3 layer with >= 32
    input channels (forward, data gradient, and the R1 double backward built from them) runs the F(2x2,3x3) kernels
    — at the default threshold the small golden resolutions would stay on the direct kernels.

    Tolerances are the direct path's, except the gradient w.r.t. the discriminator's input image: Winograd's fp32
    rounding (3e-6 of the layer's max) flips the leaky-ReLU branch of a handful of near-zero pre-activations, which
    moves the few image-gradient pixels downstream of them by ~1e-3 of the max (measured: 27 of 12288 elements;
    with a direct forward and a Winograd backward the same gradient agrees to 2e-6).  So that tensor is held to a
    relative L2 bound plus a cap on the number of outliers instead of an L_inf bound."""
    from gif_amd import losses, ops
    monkeypatch.setattr(ops, "WINOGRAD_MIN_TILES", 0)
    monkeypatch.setattr(ops, "WINOGRAD_MIN_C", 0)
    monkeypatch.setattr(ops, "WINOGRAD_WGRAD_MIN_C", 0)
    before = ops.prof_winograd_calls()
    test_generator_golden_forward_backward()
    test_generator_config1_golden()
    test_generator_and_discriminator_256_vs_oracle()
    assert ops.prof_winograd_calls() > before
    gold = _gold()
    c = gold["d32"]
    d = _build_d(32)
    d.load_state_dict(d_state(gold, c["seed"]), strict=True)
    d = d.cuda()
    img = c["img"].cuda().requires_grad_(True)
    scores, _ = d([img], condition=c["cond"].cuda())
    assert_close(scores, c["scores"], 1e-4, "D scores (winograd)")
    pen = losses.grad_penalty_loss([img], scores, step=None)
    assert_close(pen, c["r1"], 2e-4, "R1 penalty (winograd)")
    (F.softplus(-scores).mean() + pen.mean()).backward()
    assert_close(d.convs[0][0].weight.grad, c["grad_first_w"], 3e-4, "D grad first conv (winograd)")
    assert_close(d.convs[1].conv2[1].weight.grad[:4], c["grad_res1_conv2_w"], 3e-4, "D grad res1.conv2 (winograd)")
    assert_close(d.final_conv[0].weight.grad[:2], c["grad_final_conv_w"], 3e-4, "D grad final_conv (winograd)")
    assert_close(d.final_linear[1].weight.grad, c["grad_lin1_w"], 3e-4, "D grad last linear (winograd)")
    diff = (img.grad.cpu() - c["grad_img"]).abs()
    ref = c["grad_img"]
    assert (diff.norm() / ref.norm()).item() < 1e-3, "grad wrt image, relative L2"
    assert (diff > 3e-4 * ref.abs().max()).float().mean().item() < 0.01, "grad wrt image: too many outliers"
    assert diff.max().item() < 1e-2 * ref.abs().max().item()


def test_path_length_regulariser_value_parity_with_reference_class(monkeypatch):
    """a13: PathLengthRegularizor(reference_semantics=True) against penalties / moving means produced by the REAL reference
    class on a stand-in generator, fed the very same random draws (tests/golden/make_pathlen_golden.py)."""
    import numpy as np
    from golden.make_pathlen_golden import standin_generator, standin_weights
    from gif_amd import losses
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pathlen_golden.npz"))
    draws = [torch.from_numpy(g[f"draw{i}"]) for i in range(4)]
    A, bias = standin_weights()
    gen = standin_generator(A.cuda(), bias.cuda())

    def replay(*a, **k):
        t = draws.pop(0).to(k.get("device", "cpu"))
        assert tuple(t.shape) == tuple(a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a)
        return t.requires_grad_(k.get("requires_grad", False))

    monkeypatch.setattr(torch, "randn", replay)
    reg = losses.PathLengthRegularizor(reference_semantics=True)
    idx = torch.tensor([0, 1, 2], device="cuda")
    for k in range(2):
        pen = reg.path_length_reg(gen, 1, 1.0, idx)
        assert abs(pen.item() - g["penalty"][k]) < 1e-5 * g["penalty"][k], (k, pen.item(), g["penalty"][k])
        assert abs(float(reg.pl_moving_mean) - g["moving_mean"][k]) < 1e-5 * g["moving_mean"][k]
        assert not pen.requires_grad, "reference semantics: no create_graph => the penalty carries no gradient"


def test_training_trajectory_vs_oracle_trainer():
    """a14 end to end: two full iterations (the second an R1 iteration) of gif_amd.train_step.GifTrainer against the CPU
    restatement of train.py:82-250 (oracle/train_ref.py) from the same seeded weights and the same batches: losses of both
    iterations, parameters after the Adam steps and the EMA generator.

    Adam with beta1 = 0 moves every weight by ~lr * sign(grad) on its first step, so an entry whose gradient is within rounding
    of zero may move the other way: parameters are compared on the fraction of entries that agree, losses on their values."""
    from oracle import stylegan2_ref as R
    from oracle.train_ref import RefTrainer
    from gif_amd.train_step import GifTrainer
    torch.manual_seed(0)
    G, G_ema, D = _build_g(vocab=16), _build_g(vocab=16), _build_d(32)
    g_sd = R.seeded_state_dict(G.state_dict(), 61)
    d_sd = R.seeded_state_dict(D.state_dict(), 62)
    G.load_state_dict(g_sd, strict=True)
    G_ema.load_state_dict(g_sd, strict=True)
    D.load_state_dict(d_sd, strict=True)
    ref = RefTrainer(g_sd, d_sd, res_step=3, size=32, r1_every=2)
    tr = GifTrainer(G.cuda(), D.cuda(), G_ema.cuda(), step=3, alpha=1.0, r1_every=2)
    gen = torch.Generator().manual_seed(63)
    for i in range(2):
        real = torch.rand(4, 3, 32, 32, generator=gen) * 2 - 1
        cond = torch.rand(4, 6, 32, 32, generator=gen) * 2 - 1
        idx = torch.randint(0, 16, (4,), generator=gen)
        d_ref, g_ref = ref.step(i, real, cond, idx)
        d_got, g_got = tr.step(i, real.cuda(), cond.cuda(), idx.cuda())
        assert abs(d_got.item() - d_ref.item()) < 2e-3 * max(1.0, abs(d_ref.item())), (i, d_got.item(), d_ref.item())
        assert abs(g_got.item() - g_ref.item()) < 2e-3 * max(1.0, abs(g_ref.item())), (i, g_got.item(), g_ref.item())
    lr_g, lr_d = 0.002 * 4 / 5, 0.002 * 16 / 17
    for model, leaves, lr in ((G, ref.g, lr_g), (D, ref.d, lr_d)):
        sd = model.state_dict()
        agree, total = 0, 0
        named = dict(model.named_parameters())
        for k, v in leaves.items():
            if not v.requires_grad or named[k].grad is None:  # blocks above the trained resolution never move on either side
                continue
            diff = (sd[k].detach().cpu() - v.detach()).abs()
            agree += int((diff <= 0.05 * lr).sum())
            total += diff.numel()
            assert diff.max().item() <= 4.2 * lr, (k, diff.max().item())  # at most two sign flips of lr-sized steps
        print(f"trajectory agreement: {agree / total:.4f} ({total} weights)")
        assert agree / total > 0.985, (agree, total)  # (0.992 observed: 0.8 % of the ~21 M trained weights have |grad| ~ rounding)
    decay = 0.5 ** (32 / 10000)
    w_ema = G_ema.state_dict()["generator.progression.2.st_cv2.conv.weight"].cpu()
    assert (w_ema - ref.g_ema["generator.progression.2.st_cv2.conv.weight"]).abs().max().item() < 4.2 * lr_g * (1 - decay) * 2 + 1e-6


def test_full_batch_256_matches_the_oracle_checked_small_batch():
    """BASELINE size (256x256, batch 32): samples are independent in G (and in D up to the stddev groups of 4), so the
    matching samples of a batch-32 run must equal a batch-4 run — which is the configuration checked against the CPU oracle above.  The
    two runs take different kernels (Winograd tile threshold, tile shapes, launch splits), so this ties the full-size
    dispatch to the oracle-checked one."""
    from oracle import stylegan2_ref as R
    torch.manual_seed(0)
    g = _build_g(vocab=64)
    g.load_state_dict(R.seeded_state_dict(g.state_dict(), 21), strict=True)
    d = _build_d(256)
    d.load_state_dict(R.seeded_state_dict(d.state_dict(), 22), strict=True)
    g, d = g.cuda().eval(), d.cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(5)
    cond = torch.rand(32, 6, 256, 256, device="cuda", generator=gen) * 2 - 1
    idx = torch.randint(0, 64, (32,), device="cuda", generator=gen)
    with torch.no_grad():
        big = g(cond, None, step=6, alpha=1, input_indices=idx)[0]
        small = g(cond[:4], None, step=6, alpha=1, input_indices=idx[:4])[0]
        assert_close(big[:4], small.cpu(), 1e-4, "G: batch 32 vs batch 4")
        assert (big[:4] - small).abs().max().item() < 1e-3
        # minibatch stddev: view(group=4, B/4, ...) puts samples {b, b+8, b+16, b+24} of a batch of 32 into one group
        # (stg2_discriminator.py:59-65), so the batch-4 run that shares sample 0's statistics is exactly those four
        sel = torch.tensor([0, 8, 16, 24], device="cuda")
        s_big = d(big, condition=cond)[0]
        s_small = d(big[sel].contiguous(), condition=cond[sel].contiguous())[0]
        assert_close(s_big[sel], s_small.cpu(), 2e-4, "D: batch 32 vs its stddev group run alone")
