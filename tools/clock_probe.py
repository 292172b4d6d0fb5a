"""Samples rocm-smi (sclk, power) while one conv implementation runs in a loop: is the Winograd GEMM clock/power limited?
Usage: python tools/clock_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gif_amd import ops  # noqa: E402

ops.WINOGRAD_MIN_C = ops.WINOGRAD_WGRAD_MIN_C = 0  # (round 6: Winograd from 256 channels in f16x2 mode by default; this tool pins the route itself)


def sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip())
        except Exception as e:  # noqa: BLE001
            out.append(f"ERR {e}")
        time.sleep(0.25)


def run(label, fn, secs=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out))
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop.set()
    th.join()
    print(f"== {label}: {dt / n * 1e3:.3f} ms per call")
    for s in out[2:8]:
        print("   ", s[:400])


def main():
    B = 32
    spec = ops.ConvSpec(3, 3, 1, 1)
    x = torch.randn(B, 512, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(512, 512, 3, 3, device="cuda") / 50
    ops.WINOGRAD = False
    run("direct 512->512 @64^2", lambda: ops.conv_fwd(x, w, spec))
    ops.WINOGRAD, ops.WINOGRAD_MIN_TILES = True, 0
    run("winograd 512->512 @64^2", lambda: ops.conv_fwd(x, w, spec))
    y = torch.empty_like(x)
    run("HBM copy 0.27 GB", lambda: y.copy_(x))


if __name__ == "__main__":
    main()
