"""Which launches of a real training iteration take the f16x2 guard's bf16x3 fallback?  Wraps the conv entry points of gif_amd.ops,
reads the device's fallback counter after every call (synchronising: diagnostic only) and prints the shapes that raised it."""
import collections
import contextlib
import io
import sys

import torch

sys.path.insert(0, ".")
from gif_amd import ops  # noqa: E402
from gif_amd.discriminator import Discriminator  # noqa: E402
from gif_amd.generator import StyledGenerator  # noqa: E402
from gif_amd.train_step import GifTrainer  # noqa: E402

dev = torch.device("cuda", 0)
hits = collections.Counter()
calls = collections.Counter()


def wrap(name):
    fn = getattr(ops, name)

    def w(*a, **k):
        n0 = ops.h2_fallback_stats()
        out = fn(*a, **k)
        n1 = ops.h2_fallback_stats()
        x = a[0]
        wt = a[1]
        key = (name, tuple(x.shape), tuple(wt.shape), "mod" if k.get("in_scale") is not None else "plain")
        calls[key] += 1
        if n1 > n0:
            hits[key] += n1 - n0
            if hits[key] <= 1:
                xs = x.float()
                rowmax = xs.abs().amax(dim=1)
                print(f"FALLBACK {key}: input absmax {float(xs.abs().max()):.3e}, min non-zero row max {float(rowmax[rowmax > 0].min()):.3e}, "
                      f"zero fraction {float((xs == 0).float().mean()):.3f}", flush=True)
        return out
    setattr(ops, name, w)


for n in ("conv3x3_winograd", "conv_fwd", "conv_bwd_data"):
    wrap(n)
# conv_fwd / conv_bwd_data call conv3x3_winograd internally: the inner wrapper attributes the hit, the outer sees the same increment
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    kw = dict(embedding_vocab_size=1000, rendered_flame_ascondition=True, normal_maps_as_cond=True, core_tensor_res=4, n_mlp=8)
    G, Ge = StyledGenerator(**kw), StyledGenerator(**kw)
    D = Discriminator(size=256, num_color_chnls=9, channel_multiplier=2)
Ge.load_state_dict(G.state_dict())
G, Ge, D = G.to(dev), Ge.to(dev), D.to(dev)
tr = GifTrainer(G, D, Ge, step=6, alpha=1.0, r1_every=16)
from gif_amd.data import SyntheticBatches  # noqa: E402
bt = SyntheticBatches(8, 256, 1000, dev, seed=1, rank=0)
for it in range(3):
    real, cond, idx = next(bt)
    tr.step(it, real, cond, idx)
    print(f"--- iteration {it}: fallbacks so far {ops.h2_fallback_stats()}", flush=True)
tr.flush()
print("\nper launch shape (hits / calls):")
for k, v in sorted(hits.items(), key=lambda kv: -kv[1]):
    print(f"  {v:4d} / {calls[k]:4d}  {k}")
