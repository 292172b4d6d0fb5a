"""Generates tests/golden/stylegan2_golden.pt by running the REAL reference modules (imported in place from
/root/reference through oracle/reference_import.py — build container only).

    python tests/golden/make_stylegan2_golden.py

Weights are not stored: they are regenerated from oracle.stylegan2_ref.seeded_state_dict(template, seed), whose
only input is the key->shape template (recorded here) and a seed.  Stored: inputs, reference outputs, a few
reference gradients, all small (32x32 / 64x64 configs incl. BASELINE config 1's shape: step=4, 64x64, batch 4,
zero condition, float z).
"""
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_import as ri  # noqa: E402
from oracle import stylegan2_ref as R  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stylegan2_golden.pt")


def main():
    SG, D, L = ri.reference_modules()
    with contextlib.redirect_stdout(io.StringIO()):
        g = SG(embedding_vocab_size=50, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        d = D(size=32, num_color_chnls=9)
    gold = {"g_template": {k: tuple(v.shape) for k, v in g.state_dict().items()},
            "d_template": {k: tuple(v.shape) for k, v in d.state_dict().items()},
            "g_kernels": {k: v.clone() for k, v in g.state_dict().items() if k.endswith(".kernel")},
            "d_kernels": {k: v.clone() for k, v in d.state_dict().items() if k.endswith(".kernel")}}
    gen = torch.Generator().manual_seed(1234)

    # --- generator, step 3 (32x32), batch 2, embedding indices
    sd = R.seeded_state_dict(g.state_dict(), 11)
    g.load_state_dict(sd, strict=True)
    cond = torch.rand(2, 6, 32, 32, generator=gen) * 2 - 1
    idx = torch.tensor([3, 41])
    out = g(cond, None, step=3, alpha=1, input_indices=idx)[0]
    loss = (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum()
    g.zero_grad()
    loss.backward()
    gold["g32"] = {"seed": 11, "cond": cond, "idx": idx, "out": out.detach(),
                   "grad_const": g.generator.const_input.input.grad.clone(),
                   "grad_w_4x4": g.generator.progression[0].st_cv1.conv.weight.grad[0, :2].clone(),
                   "grad_mod_b_16": g.generator.progression[2].st_cv1.conv.modulation.bias.grad.clone(),
                   "grad_noise_w_32": g.generator.progression[3].st_cv2.noise.noise_conv[4].weight.grad[:8].clone(),
                   "grad_rgb_w_32": g.generator.to_rgb[3].conv.weight.grad.clone(),
                   "grad_z_to_w_8_b": g.z_to_w[8].bias.grad.clone()}

    # --- BASELINE config 1 shape: step 4 (64x64), batch 4, zero condition, float z
    z = torch.randn(4, 512, generator=gen)
    with torch.no_grad():
        out64 = g(torch.zeros(4, 6, 64, 64), None, step=4, alpha=1, input_indices=z)[0]
    gold["g64"] = {"seed": 11, "z": z, "out": out64}

    # --- discriminator 32x32, batch 4 (stddev group 4), scores, R1 penalty and its parameter gradients
    sdd = R.seeded_state_dict(d.state_dict(), 12)
    d.load_state_dict(sdd, strict=True)
    img = (torch.rand(4, 3, 32, 32, generator=gen) * 2 - 1).requires_grad_(True)
    c2 = torch.rand(4, 6, 32, 32, generator=gen) * 2 - 1
    scores = d([img], condition=c2)[0]
    pen = R.grad_penalty_loss([img], scores)  # same arithmetic as losses.py:87-99 (losses.py itself is not importable)
    total = torch.nn.functional.softplus(-scores).mean() + pen.mean()
    d.zero_grad()
    total.backward()
    gold["d32"] = {"seed": 12, "img": img.detach(), "cond": c2, "scores": scores.detach(), "r1": pen.detach(),
                   "grad_first_w": d.convs[0][0].weight.grad.clone(),
                   "grad_res1_conv2_w": d.convs[1].conv2[1].weight.grad[:4].clone(),
                   "grad_final_conv_w": d.final_conv[0].weight.grad[:2].clone(),
                   "grad_lin1_w": d.final_linear[1].weight.grad.clone(),
                   "grad_img": img.grad.clone()}
    # --- batch 6: group = min(6,4) = 4 does not divide 6 -> the reference raises; batch 8 -> two groups
    img8 = torch.rand(8, 3, 32, 32, generator=gen) * 2 - 1
    c8 = torch.rand(8, 6, 32, 32, generator=gen) * 2 - 1
    with torch.no_grad():
        gold["d32_b8"] = {"img": img8, "cond": c8, "scores": d([img8], condition=c8)[0]}
    torch.save(gold, OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
