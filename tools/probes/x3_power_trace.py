"""rocm-smi power / sclk trace taken DURING the bf16x3 kernels (verdict r2 item 4: the power-limited reading of §3a rested on a
zero-data probe and PMC-derived clocks only).  For each kernel: loop it for ~3 s, sample `rocm-smi --showpower --showclocks`
every 0.2 s from a side thread, print per-launch time, algorithmic TFLOP/s and the samples' mean / max.  Each kernel runs on
random data and on all-zero data (same instruction stream, no toggling in the MFMA datapath).
Usage (GPU): python tools/probes/x3_power_trace.py > profiles/r3_x3_power_trace.txt
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)


def sample():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
        card = next(iter(json.loads(r.stdout).values()))
        power = sclk = None
        for k, v in card.items():
            if "power" in k.lower() and power is None:
                m = re.search(r"[\d.]+", str(v))
                power = float(m.group()) if m else None
            if k.lower().startswith("sclk") and "speed" in k.lower():
                m = re.search(r"(\d+)\s*mhz", str(v).lower())
                sclk = float(m.group(1)) if m else sclk
        return power, sclk, None
    except Exception as e:  # noqa: BLE001
        return None, None, repr(e)


def run(label, fn, flops, secs=3.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []

    def loop():
        while not stop.is_set():
            out.append(sample())
            time.sleep(0.2)
    th = threading.Thread(target=loop)
    th.start()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop.set()
    th.join()
    pw = [p for p, _, _ in out[2:] if p is not None]
    ck = [c for _, c, _ in out[2:] if c is not None]
    err = [e for _, _, e in out if e]
    ms = dt / n * 1e3
    print(f"{label:46s} {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF alg | power W mean {sum(pw) / max(len(pw), 1):7.1f} max {max(pw, default=0):7.1f} | "
          f"sclk MHz mean {sum(ck) / max(len(ck), 1):6.0f} min {min(ck, default=0):6.0f} | {len(pw)} samples" + (f" | {err[0]}" if err else ""), flush=True)


def main():
    B, dev = 32, "cuda"
    print("# " + " ".join(sys.argv), "| mode", ops.get_fp32_mfma_mode())
    spec = ops.ConvSpec(3, 3, 1, 1)
    for zero in (False, True):
        mk = (lambda *s: torch.zeros(*s, device=dev)) if zero else (lambda *s: torch.randn(*s, device=dev))
        tag = "zeros " if zero else "randn "
        x = mk(B, 512, 64, 64).contiguous(memory_format=torch.channels_last)
        w = mk(512, 512, 3, 3) / 50
        fl = 2.0 * B * 64 * 64 * 9 * 512 * 512
        ops.WINOGRAD = False
        run(tag + "direct bf16x3 conv 512->512 @64^2", lambda: ops.conv_fwd(x, w, spec), fl)
        run(tag + "direct bf16x3 wgrad 512x512 @64^2", lambda: ops.conv_wgrad(x, x, spec, 512, 512), fl)
        ops.WINOGRAD = True
        run(tag + "Winograd bf16x3 conv 512->512 @64^2 (+transform)", lambda: ops.conv_fwd(x, w, spec), fl)
        run(tag + "Winograd bf16x3 wgrad 512x512 @64^2 (+transforms)", lambda: ops.conv_wgrad(x, x, spec, 512, 512), fl)
        x2 = mk(B, 128, 256, 256).contiguous(memory_format=torch.channels_last)
        w2 = mk(128, 128, 3, 3) / 30
        fl2 = 2.0 * B * 256 * 256 * 9 * 128 * 128
        run(tag + "Winograd bf16x3 conv 128->128 @256^2 (+transform)", lambda: ops.conv_fwd(x2, w2, spec), fl2)
        ops.WINOGRAD = False
        run(tag + "direct bf16x3 conv 128->128 @256^2", lambda: ops.conv_fwd(x2, w2, spec), fl2)
        del x2
    ops.set_fp32_mfma_mode("native")
    x = torch.randn(B, 512, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(512, 512, 3, 3, device=dev) / 50
    run("randn direct native fp32-MFMA conv 512->512 @64^2", lambda: ops.conv_fwd(x, w, spec), 2.0 * B * 64 * 64 * 9 * 512 * 512)
    y = torch.empty_like(x)
    run("HBM copy 0.27 GB", lambda: y.copy_(x), 0.0)


if __name__ == "__main__":
    main()
