"""Checkpoint compatibility with the reference's training script.

train.py:254-265 saves, every 1000 iterations,
    torch.save({'generator_running', 'generator', 'g_optimizer', 'discriminator_flm', 'd_optimizer_flm'}, '<it>_<alpha>.model')
    np.savez('<it>_<alpha>.npz', step=, used_sampless=, alpha=, resolution=)
where the three networks are nn.DataParallel wrappers, so every state_dict key carries a 'module.' prefix, and
plots/generate_random_samples.py:143-144 loads ckpt['generator_running'] into a DataParallel-wrapped generator.
These helpers write / read exactly that format from the un-wrapped per-rank modules of GifTrainer (rank 0 saves).
"""
import os

import numpy as np
import torch

_NETS = ("generator_running", "generator", "discriminator_flm")


def add_module_prefix(state_dict):
    return {("module." + k if not k.startswith("module.") else k): v for k, v in state_dict.items()}


def strip_module_prefix(state_dict):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def checkpoint_dict(trainer):
    """The reference's 5-entry checkpoint dict, DataParallel-style keys (train.py:257-261)."""
    if hasattr(trainer, "flush"):
        trainer.flush()  # complete a discriminator update deferred behind the generator forward (overlap_comm)
    return {"generator_running": add_module_prefix(trainer.G_ema.state_dict()),
            "generator": add_module_prefix(trainer.G.state_dict()),
            "g_optimizer": trainer.g_optim.state_dict(),
            "discriminator_flm": add_module_prefix(trainer.D.state_dict()),
            "d_optimizer_flm": trainer.d_optim.state_dict()}


def save_checkpoint(trainer, path, step, used_samples, alpha=1.0, resolution=None, rank=0):
    """Writes '<path>' (.model) and the side-car .npz of train.py:263-265.  Only rank 0 writes."""
    if rank != 0:
        return None
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(checkpoint_dict(trainer), path)
    np.savez(path.replace(".model", ".npz"), step=step, used_sampless=used_samples, alpha=alpha,
             resolution=resolution if resolution is not None else 4 * 2 ** step)
    return path


def load_checkpoint(trainer, path, map_location=None, strict=True):
    """Restores the five entries (train.py:389-400); accepts keys with or without the 'module.' prefix, i.e. both
    checkpoints written here and checkpoints released with the reference.  Returns (step, used_samples) when the
    side-car .npz exists, else None."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    trainer.G.load_state_dict(strip_module_prefix(ckpt["generator"]), strict=strict)
    trainer.G_ema.load_state_dict(strip_module_prefix(ckpt["generator_running"]), strict=strict)
    trainer.D.load_state_dict(strip_module_prefix(ckpt["discriminator_flm"]), strict=strict)
    trainer.g_optim.load_state_dict(ckpt["g_optimizer"])
    trainer.d_optim.load_state_dict(ckpt["d_optimizer_flm"])
    npz = path.replace(".model", ".npz")
    if os.path.exists(npz):
        v = np.load(npz)
        return int(v["step"]), int(v["used_sampless"])
    return None


def load_generator_for_inference(generator, path_or_ckpt, key="generator_running", map_location=None):
    """plots/generate_random_samples.py:143-145: load the EMA generator of a (reference or gif_amd) checkpoint."""
    ckpt = path_or_ckpt if isinstance(path_or_ckpt, dict) else torch.load(path_or_ckpt, map_location=map_location,
                                                                           weights_only=False)
    generator.load_state_dict(strip_module_prefix(ckpt[key]), strict=True)
    return generator.eval()
