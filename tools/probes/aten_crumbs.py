"""GPU probe: which Python call sites launch the ATen glue kernels of a training iteration (add / copy / fill / mul ...), with tensor
shapes: torch.profiler over two iterations, grouped by (op, input shapes, innermost gif_amd / bench frame).
python tools/probes/aten_crumbs.py [--dtype f16] [--res 256] [--batch 32]"""
import argparse
import collections
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    from gif_amd.data import SyntheticBatches
    from gif_amd.discriminator import Discriminator
    from gif_amd.generator import StyledGenerator
    from gif_amd.train_step import GifTrainer
    dev = torch.device("cuda")
    step = {64: 4, 128: 5, 256: 6, 512: 7, 1024: 8}[a.res]
    with contextlib.redirect_stdout(io.StringIO()):
        kw = dict(embedding_vocab_size=1024, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        G, Ge, D = StyledGenerator(**kw).to(dev), StyledGenerator(**kw).to(dev), Discriminator(size=a.res, num_color_chnls=9).to(dev)
    tr = GifTrainer(G, D, Ge, step=step, act_dtype=torch.float16 if a.dtype == "f16" else None)
    batches = SyntheticBatches(a.batch, a.res, 1024, dev, seed=1, rank=0)
    for i in range(2):
        tr.step(i, *next(batches))
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for i in range(2, 4):
            tr.step(i, *next(batches))
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        us = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if not ev.key.startswith("aten::") or us <= 0:
            continue
        frame = "?"
        for fr in ev.stack or []:
            if "gif_amd" in fr or "bench.py" in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        rows.append((us, ev.count, ev.key, str(ev.input_shapes)[:80], frame[:120]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"ATen ops with self device time over 2 iterations ({a.dtype}, {a.res}x{a.res}, batch {a.batch}): {sum(r[1] for r in rows)} calls, {tot / 2e3:.2f} ms per iteration")
    for us, n, name, shapes, frame in rows[:70]:
        print(f"{n / 2:7.1f}/it {us / 2e3:8.3f} ms/it  {name:26s} {shapes:80s} {frame}")


if __name__ == "__main__":
    main()
