"""Checkpoint format compatibility (train.py:254-265, :389-400) — CPU only: the modules are built and their
state_dicts round-tripped; no kernel runs."""
import contextlib
import io
import os

import numpy as np
import torch

from gif_amd import checkpoint as C
from gif_amd.discriminator import Discriminator
from gif_amd.generator import StyledGenerator
from gif_amd.train_step import GifTrainer


def _nets():
    with contextlib.redirect_stdout(io.StringIO()):
        kw = dict(embedding_vocab_size=8, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        return StyledGenerator(**kw), StyledGenerator(**kw), Discriminator(size=16, num_color_chnls=9)


def test_checkpoint_round_trip_and_reference_key_format(tmp_path):
    torch.manual_seed(0)
    g, ge, d = _nets()
    tr = GifTrainer(g, d, ge, step=2, fused_adam=False)
    ck = C.checkpoint_dict(tr)
    assert set(ck) == {"generator_running", "generator", "g_optimizer", "discriminator_flm", "d_optimizer_flm"}
    assert all(k.startswith("module.") for k in ck["generator"]) and "module.generator.const_input.input" in ck["generator"]
    assert "module.img_embdng.embd_weight" in ck["generator_running"]  # both embedding keys, like the reference
    path = str(tmp_path / "checkpoint" / "29" / "001000_1.model")
    C.save_checkpoint(tr, path, step=2, used_samples=16000)
    assert os.path.exists(path.replace(".model", ".npz"))
    v = np.load(path.replace(".model", ".npz"))
    assert set(v.files) == {"step", "used_sampless", "alpha", "resolution"} and int(v["resolution"]) == 16
    torch.manual_seed(1)
    g2, ge2, d2 = _nets()
    tr2 = GifTrainer(g2, d2, ge2, step=2, fused_adam=False)
    assert not torch.equal(g2.generator.const_input.input, g.generator.const_input.input)
    assert C.load_checkpoint(tr2, path) == (2, 16000)
    for a, b in ((g, g2), (ge, ge2), (d, d2)):
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
    # a reference-style checkpoint WITHOUT our helper: plain dict of prefixed tensors loads into a bare generator
    g3, _, _ = _nets()
    C.load_generator_for_inference(g3, {"generator_running": C.add_module_prefix(ge.state_dict())})
    assert torch.equal(g3.z_to_w[1].weight, ge.z_to_w[1].weight) and not g3.training
