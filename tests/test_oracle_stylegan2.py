"""Pins oracle/stylegan2_ref.py: (a) against golden vectors produced by the real reference
(tests/golden/make_stylegan2_golden.py) — runs everywhere; (b) against the real reference modules imported in
place — build container only (marker `reference`)."""
import contextlib
import io
import os

import pytest
import torch

from oracle import stylegan2_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "stylegan2_golden.pt")


def _gold():
    return torch.load(GOLD, weights_only=True)


def _template(shapes, kernels):
    t = {k: torch.zeros(s) for k, s in shapes.items()}
    t.update(kernels)
    return t


def g_state(gold, seed):
    return R.seeded_state_dict(_template(gold["g_template"], gold["g_kernels"]), seed)


def d_state(gold, seed):
    return R.seeded_state_dict(_template(gold["d_template"], gold["d_kernels"]), seed)


def test_generator_matches_reference_golden():
    gold = _gold()
    c = gold["g32"]
    sd = {k: v.requires_grad_(True) if v.is_floating_point() and not k.endswith("kernel") and "embd" not in k else v
          for k, v in g_state(gold, c["seed"]).items()}
    out = R.generator_forward(sd, c["cond"], 3, c["idx"])
    assert torch.equal(out.detach(), c["out"]) or (out.detach() - c["out"]).abs().max() < 1e-5
    loss = (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum()
    loss.backward()
    pairs = [("generator.const_input.input", c["grad_const"], slice(None)),
             ("generator.progression.2.st_cv1.conv.modulation.bias", c["grad_mod_b_16"], slice(None)),
             ("generator.to_rgb.3.conv.weight", c["grad_rgb_w_32"], slice(None)),
             ("z_to_w.8.bias", c["grad_z_to_w_8_b"], slice(None))]
    for key, ref, sl in pairs:
        got = sd[key].grad[sl]
        assert (got - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-6, key
    assert (sd["generator.progression.0.st_cv1.conv.weight"].grad[0, :2] - c["grad_w_4x4"]).abs().max() < 1e-4
    assert (sd["generator.progression.3.st_cv2.noise.noise_conv.4.weight"].grad[:8] - c["grad_noise_w_32"]).abs().max() \
        <= 1e-4 * c["grad_noise_w_32"].abs().max()


def test_generator_config1_shape_golden():
    gold = _gold()
    c = gold["g64"]
    with torch.no_grad():
        out = R.generator_forward(g_state(gold, c["seed"]), torch.zeros(4, 6, 64, 64), 4, c["z"])
    assert out.shape == (4, 3, 64, 64)
    assert (out - c["out"]).abs().max() < 1e-5


def test_discriminator_r1_matches_reference_golden():
    gold = _gold()
    c = gold["d32"]
    sd = {k: (v.requires_grad_(True) if not k.endswith("kernel") else v) for k, v in d_state(gold, c["seed"]).items()}
    img = c["img"].clone().requires_grad_(True)
    scores = R.discriminator_forward(sd, img, c["cond"], 32)
    assert (scores.detach() - c["scores"]).abs().max() < 1e-6
    pen = R.grad_penalty_loss([img], scores)
    assert (pen.detach() - c["r1"]).abs().max() <= 1e-5 * c["r1"].abs().max()
    (torch.nn.functional.softplus(-scores).mean() + pen.mean()).backward()
    assert (sd["convs.0.0.weight"].grad - c["grad_first_w"]).abs().max() <= 1e-4 * c["grad_first_w"].abs().max()
    assert (sd["final_linear.1.weight"].grad - c["grad_lin1_w"]).abs().max() <= 1e-4 * c["grad_lin1_w"].abs().max()
    assert (img.grad - c["grad_img"]).abs().max() <= 1e-4 * c["grad_img"].abs().max()
    with torch.no_grad():
        s8 = R.discriminator_forward(d_state(gold, c["seed"]), gold["d32_b8"]["img"], gold["d32_b8"]["cond"], 32)
    assert (s8 - gold["d32_b8"]["scores"]).abs().max() < 1e-6


def test_upfirdn2d_shapes_and_edge_cases():
    k = R.make_kernel([1, 3, 3, 1])
    x = torch.randn(2, 3, 8, 8)
    assert R.upfirdn2d(x, k * 4, up=2, pad=(2, 1)).shape == (2, 3, 16, 16)   # Upsample
    assert R.upfirdn2d(x, k, pad=(2, 2)).shape == (2, 3, 9, 9)               # D blur before stride-2 3x3
    assert R.upfirdn2d(x, k, pad=(1, 1)).shape == (2, 3, 7, 7)               # D blur before stride-2 1x1
    assert R.upfirdn2d(x, k, down=2, pad=(1, 1)).shape == (2, 3, 4, 4)       # Downsample
    assert R.upfirdn2d(x, k, pad=(-1, 2)).shape == (2, 3, 6, 6)              # negative pad = crop
    # DC gain 1 for the normalised kernel in the interior
    ones = torch.ones(1, 1, 8, 8)
    assert torch.allclose(R.upfirdn2d(ones, k, pad=(2, 2))[0, 0, 3:6, 3:6], torch.ones(3, 3), atol=1e-6)


@pytest.mark.reference
def test_oracle_equals_imported_reference():
    from oracle import reference_import as ri
    SG, D, L = ri.reference_modules()
    with contextlib.redirect_stdout(io.StringIO()):
        g = SG(embedding_vocab_size=20, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        d = D(size=64, num_color_chnls=9)
    sd = R.seeded_state_dict(g.state_dict(), 5)
    g.load_state_dict(sd, strict=True)
    cond = torch.rand(2, 6, 64, 64) * 2 - 1
    with torch.no_grad():
        for idx in (torch.tensor([1, 19]), torch.randn(2, 512)):
            ref = g(cond, None, step=4, alpha=1, input_indices=idx)[0]
            assert (R.generator_forward(sd, cond, 4, idx) - ref).abs().max() < 1e-6
    sdd = R.seeded_state_dict(d.state_dict(), 6)
    d.load_state_dict(sdd, strict=True)
    img = torch.rand(4, 3, 64, 64) * 2 - 1
    c2 = torch.rand(4, 6, 64, 64) * 2 - 1
    with torch.no_grad():
        assert (R.discriminator_forward(sdd, img, c2, 64) - d(img, condition=c2)[0]).abs().max() < 1e-6
    # layer-level: upfirdn2d incl. negative pads, FusedLeakyReLU
    k = L.make_kernel([1, 3, 3, 1])
    x = torch.randn(2, 5, 9, 9)
    for up, down, pad in ((1, 1, (2, 2)), (2, 1, (2, 1)), (1, 2, (1, 1)), (1, 1, (-1, 2)), (2, 2, (3, 0))):
        assert torch.equal(R.upfirdn2d(x, k, up, down, pad), L.upfirdn2d(x, k, up, down, pad))


@pytest.mark.reference
def test_oracle_layer_variants_equal_imported_reference():
    """Less-travelled layer options: down-sampling modulated conv, no-demod 1x1 (ToRGB), up-sampling modulated conv."""
    from oracle import reference_import as ri
    _, _, L = ri.reference_modules()
    torch.manual_seed(3)
    x, st = torch.randn(2, 16, 8, 8), torch.randn(2, 512)
    for kw in (dict(downsample=True), dict(upsample=True), dict()):
        m = L.ModulatedConv2d(16, 24, 3, 512, **kw)
        sd = m.state_dict()
        ref = m(x, st)
        got = R.modulated_conv2d(x, sd["weight"], sd["modulation.weight"], sd["modulation.bias"], st, True,
                                 kw.get("upsample", False), sd.get("blur.kernel"), kw.get("downsample", False))
        assert ref.shape == got.shape and (ref - got).abs().max() < 1e-5, kw
    m = L.ModulatedConv2d(16, 3, 1, 512, demodulate=False)
    sd = m.state_dict()
    assert (m(x, st) - R.modulated_conv2d(x, sd["weight"], sd["modulation.weight"], sd["modulation.bias"], st,
                                          demodulate=False)).abs().max() < 1e-5
