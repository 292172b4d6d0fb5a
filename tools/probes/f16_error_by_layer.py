"""Where does the f16-activation path's output error come from?  (round-2 review: the 3e-2 bound of tests/test_gpu_f16.py was
stated without an analysis.)  Runs the generator at 256x256 in fp32 and with f16 activations on the SAME weights and inputs and
prints, per block, the error of the StyledConv outputs and of the running RGB image, each relative to its own magnitude — plus
the same with each block's f16 input REPLACED by the fp32 run's (rounded to f16): the error a block adds on its own.
Usage (GPU): python tools/probes/f16_error_by_layer.py > profiles/r3_f16_error_by_layer.txt
"""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd.generator import StyledGenerator  # noqa: E402
from oracle import stylegan2_ref as R  # noqa: E402  (seeded weights only: the same ones the parity tests use)


def run(g, cond, idx, step, dtype, taps):
    g.set_activation_dtype(dtype)
    hooks = []
    gen = g.generator
    for i in range(step + 1):
        blk = gen.progression[i]
        for name in ("st_cv1", "st_cv2"):
            if hasattr(blk, name):
                hooks.append(getattr(blk, name).register_forward_hook(
                    lambda m, inp, out, key=f"block{i}.{name}": taps.__setitem__(key, out.detach().float())))
        hooks.append(gen.to_rgb[i].register_forward_hook(lambda m, inp, out, key=f"block{i}.rgb": taps.__setitem__(key, out.detach().float())))
    with torch.no_grad():
        img = g(cond, None, step=step, alpha=1, input_indices=idx)[0]
    for h in hooks:
        h.remove()
    g.set_activation_dtype(torch.float32)
    return img


def main():
    torch.manual_seed(0)
    step, res, B = 6, 256, 4
    with contextlib.redirect_stdout(io.StringIO()):
        g = StyledGenerator(embedding_vocab_size=16, rendered_flame_ascondition=True, normal_maps_as_cond=True)
    g.load_state_dict(R.seeded_state_dict(g.state_dict(), 3))
    g = g.cuda()
    cond = (torch.rand(B, 6, res, res) * 2 - 1).cuda()
    idx = torch.tensor([1, 5, 9, 13]).cuda()
    t32, t16 = {}, {}
    img32 = run(g, cond, idx, step, torch.float32, t32)
    img16 = run(g, cond, idx, step, torch.float16, t16)
    print(f"# generator at {res}x{res}, batch {B}: f16 activations vs fp32 on the same weights")
    print(f"image L_inf {float((img16 - img32).abs().max()):.3e} (image max {float(img32.abs().max()):.2f}); half-precision unit roundoff 2^-11 = {2.0 ** -11:.2e}")
    print("| tensor | max |fp32| | L_inf error | error / max | rms error / rms |")
    print("|---|---:|---:|---:|---:|")
    for k in t32:
        a, b = t32[k], t16[k][:, :t32[k].shape[1]]
        c = min(a.shape[1], b.shape[1])
        a, b = a[:, :c], b[:, :c]
        e = (a - b).abs()
        print(f"| {k} | {float(a.abs().max()):.3f} | {float(e.max()):.3e} | {float(e.max() / a.abs().max()):.3e} | "
              f"{float(e.pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e} |")
    # the per-layer increments: error of layer k+1's output relative to rounding its own fp32 output to half
    print("\n# rounding floor: the fp32 tensors rounded to f16 (what a perfect f16 pipeline would store)")
    for k in t32:
        a = t32[k]
        e = (a - a.half().float()).abs()
        print(f"| {k} | storage rounding L_inf {float(e.max()):.3e} | / max {float(e.max() / a.abs().max()):.3e} |")


if __name__ == "__main__":
    main()
