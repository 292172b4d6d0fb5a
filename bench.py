#!/usr/bin/env python
"""bench.py — G+D train-step images/sec at 256x256, batch 32/GPU (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 16 --warmup 2
    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full GIF training iteration (train.py:82-250): D step (D fwd real, G fwd, D fwd fake, backward,
Adam) + G step (G fwd, D fwd, backward, Adam, EMA), R1 on every 16th iteration, run-29 model configuration
(6-channel rendered condition, 9-channel D input, n_mlp=8, 69 158-entry embedding buffer), synthetic 256x256
batches generated on device, random-init weights, fp32 (the reference's dtype) on the fp32 MFMA path.
Other BASELINE configs as extra lines: `--batch 16` (config 2), `--render-cond [--gen-reg PATH_LEN_REG]` (config 3: the
condition is rasterised from a posed mesh INSIDE the timed region).
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: needed by RCCL on this driver stack
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16 dense peak (not the 2:1-sparse figure)
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s is what a float4 copy reaches)
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written by tools/pmc_to_json.py from rocprofv3 --pmc passes


def r1_start_iteration(steps, warmup, r1_every):
    """Iteration index of the FIRST warm-up step.  The trainer runs R1 when (i + 1) % r1_every == 0; the counter starts so that
    round(steps / r1_every) R1 iterations fall inside the timed steps (none when that rounds to 0), and — where the iteration count
    allows it — so that the LAST warm-up step is an R1 iteration too: a warm-up has to run every path the timed region runs (the first R1
    iteration of a process allocates the double-backward buffers and loads the kernels only R1 uses: 25-105 ms once, depending on the
    box; it used to land inside the timed region)."""
    if not r1_every:
        return 0
    n_r1 = int(round(steps / r1_every))
    if n_r1 > 0:
        first = warmup + max((steps - (n_r1 - 1) * r1_every) // 2, 0)  # index of the first timed R1 iteration
        cand = warmup - 1 + r1_every
        last = warmup + steps - 1
        if warmup >= 1 and cand <= last and (last - cand) // r1_every + 1 == n_r1:
            first = cand
        return (r1_every - 1 - first) % r1_every
    return (-warmup) % r1_every  # the timed steps are iterations 1 .. steps of an R1 period: no R1 iteration among them


def kernel_source_digest():
    """sha1 over the kernel sources (gif_amd/csrc/*.hip, *.h, include/*.h): what profiles/pmc_traffic.json was measured on vs what
    this run executes (the GPU box has no .git, so a commit hash cannot be compared there)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, "gif_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "gif_amd", "csrc", "*.h"))
                       + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE: 32; config 2: 16)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--vocab", type=int, default=69158)
    ap.add_argument("--r1-every", type=int, default=16)
    ap.add_argument("--render-cond", action="store_true",
                    help="config 3: rasterise the 6-channel condition from a posed mesh inside the timed region")
    ap.add_argument("--gen-reg", type=str, default="None", help="None | PATH_LEN_REG | DIRECT_GRAD_REG (train.py:203-215)")
    ap.add_argument("--texture-interp", action="store_true",
                    help="run 29 (configurations.py:217): add the texture-space interpolation loss to every generator step "
                         "(train.py:222-238) on synthetic FLAME labels — a synthetic blend-shape mesh stands in for the FLAME layer, "
                         "a synthetic UV chart for the licensed texture space; NOT part of the headline line")
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "f16"],
                    help="activation dtype: f32 = the reference's (headline); f16 = BASELINE configs[4] (f16 activations, fp32 "
                         "weights / demodulation / accumulation, loss scaling) — use with --res 1024 --batch 8")
    ap.add_argument("--fp32-mfma", type=str, default=None, choices=["native", "bf16x3", "f16x2"],
                    help="fp32 contraction mode of the conv kernels (default: GIF_FP32_MFMA, else f16x2 — fp32 tensors, every fp32 operand "
                         "split into two f16 terms under per-row power-of-two scales, 3 f16 MFMA products, fp32 accumulation, "
                         "guarded bf16x3 fallback; bf16x3: 3 bf16 terms, 6 products; native: v_mfma_f32_32x32x2_f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4, help="CPU baseline batch (BASELINE.md §3: 4; 32 does not fit)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = calibrate: fastest of {all host cores, 64, 32, 16}")
    ap.add_argument("--cpu-timeout", type=int, default=240)
    ap.add_argument("--cpu-baseline-worker", type=str, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-prof", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--prof-every", type=int, default=4,
                    help="HIP events bracket every MFMA / transform launch of every N-th timed step (two event records per launch "
                         "cost ~4 ms per fully instrumented step: measured 231 vs 227 ms); 1 = every step")
    ap.add_argument("--reuse-generator-forward", action="store_true",
                    help="NOT the headline workload: run the generator forward once per iteration and use it for the D step "
                         "(detached) and the G step (train.py:155 and :195 call G twice on identical inputs with unchanged weights; "
                         "bit-identical losses and parameters, tests/test_gpu_models.py).  The line says so and counts one forward less.")
    ap.add_argument("--no-overlap-comm", action="store_true", help="complete each gradient exchange + optimiser step in place "
                    "(default with > 1 rank: deferred to where the network is next used)")
    ap.add_argument("--two-call-d", action="store_true",
                    help="D step as two discriminator calls (train.py:142, :169) instead of one pass over [real; fake]: with > 1 rank it "
                         "hides G's gradient exchange under D's forward on the real images (A/B for the multi-GPU run; default: fused at "
                         "every rank count, so that N = 1 and N > 1 issue the same launches)")
    ap.add_argument("--check-replicas", action="store_true",
                    help="after the run: sha1 of every rank's G / D / G_ema parameters -> `param_digest`, `replicas_identical`")
    return ap.parse_args()


def _pick_threads(threads):
    """threads == 0: BASELINE.md §3 asks for all host cores — but on a 256-thread host the oracle's grouped convolutions ran
    >10x SLOWER with 256 threads than with 16 (round-2 measurement: one batch-4 step did not finish in 420 s), so the thread
    count is calibrated on a probe (the oracle's grouped 3x3 conv 128->128 @128^2, batch 4, forward + backward) and the fastest of
    {all cores, 64, 32, 16} is used.  The choice is reported in `cores`."""
    import torch.nn.functional as F
    if threads:
        return threads, None
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (ncpu, 64, 32, 16) if c <= ncpu}, reverse=True)
    # the oracle's own convolution form: per-sample weights as one grouped conv (groups = batch, ModulatedConv2d :343-347)
    x = torch.randn(1, 4 * 128, 128, 128, requires_grad=True)
    w = torch.randn(4 * 128, 128, 3, 3, requires_grad=True)
    best, log = None, {}
    for c in cands:
        torch.set_num_threads(c)
        F.conv2d(x, w, padding=1, groups=4).sum().backward()  # warm-up of the thread pool at this size
        t0 = time.time()
        for _ in range(2):
            F.conv2d(x, w, padding=1, groups=4).sum().backward()
        log[c] = (time.time() - t0) / 2
        if best is None or log[c] < log[best]:
            best = c
    return best, log


def cpu_baseline_worker(res, step_idx, batch, threads):
    """Runs in a child process: oracle ("port") timed on the host cores — full G+D training iterations at the benchmark
    resolution on a small batch (batch 32 needs ~80 GB of activations on the CPU): a plain iteration and an R1 iteration,
    weighted 15:1 like the benchmark's R1-every-16th schedule.  Bounded: the R1 iteration is only run when the plain one
    took < 60 s; otherwise the value is the PLAIN-iteration rate alone and `kind` says so ("port, plain iterations only": an upper
    bound of the CPU rate, nothing is extrapolated)."""
    threads, calib = _pick_threads(threads)
    torch.set_num_threads(threads)
    from oracle import stylegan2_ref as R
    from oracle.train_ref import RefTrainer
    from gif_amd.discriminator import Discriminator
    from gif_amd.generator import StyledGenerator
    with contextlib.redirect_stdout(io.StringIO()):
        g = StyledGenerator(embedding_vocab_size=64, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        d = Discriminator(size=res, num_color_chnls=9)
    tr = RefTrainer(R.seeded_state_dict(g.state_dict(), 1), R.seeded_state_dict(d.state_dict(), 2), res_step=step_idx,
                    size=res, r1_every=2)
    gen = torch.Generator().manual_seed(0)
    real = torch.rand(batch, 3, res, res, generator=gen) * 2 - 1
    cond = torch.rand(batch, 6, res, res, generator=gen) * 2 - 1
    idx = torch.randint(0, 64, (batch,), generator=gen)
    t0 = time.time()
    tr.step(0, real, cond, idx)  # i = 0: no R1
    t_plain = time.time() - t0
    if t_plain < 60:
        t0 = time.time()
        tr.step(1, real, cond, idx)  # i = 1: R1 iteration (r1_every = 2 here)
        t_r1 = time.time() - t0
        r1_note = f"R1 iteration {t_r1:.1f} s"
        kind = "port"
        per_step = (15 * t_plain + t_r1) / 16
    else:
        t_r1 = None
        r1_note = "R1 iteration not run (plain iteration took >= 60 s): value = plain-iteration rate, an upper bound"
        kind = "port, plain iterations only"
        per_step = t_plain
    print(json.dumps({"value": batch / per_step, "unit": "images/s", "cores": threads, "kind": kind,
                      "sample": f"full G+D train steps at {res}x{res}, batch {batch}, oracle/train_ref.py (torch CPU fp32, "
                                f"{threads} threads of {os.cpu_count()} host cores"
                                + (f", fastest of a thread-count probe {calib}" if calib else "")
                                + f"): plain {t_plain:.1f} s, {r1_note}; value = batch / ((15*plain + R1)/16) when both ran",
                      "plain_step_s": t_plain, "r1_step_s": t_r1}))


def cpu_baseline(res, step_idx, batch, threads, timeout_s):
    """Bounded: a child process with a fixed thread count and a hard timeout, so bench.py always finishes in minutes."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", f"{res},{step_idx},{batch},{threads}"]
    env = dict(os.environ)
    if threads:
        env.update(OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # timeout or failure: report it, never block the GPU numbers
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": f"cpu baseline did not finish within {timeout_s}s ({type(e).__name__})"}


WINOGRAD_EXECUTED = 16.0 / 36.0
BF16X3_EXECUTED = 6.0  # bf16 MFMA FLOPs the bf16x3 kernels execute per algorithmic fp32 FLOP
F16X2_EXECUTED = 3.0   # f16 MFMA FLOPs the f16x2 kernels execute per algorithmic fp32 FLOP (hi*hi + hi*lo + lo*hi)

# family id -> (description, executed MFMA fraction of the algorithmic FLOPs, key in profiles/pmc_traffic.json)
FAMILIES = {
    0: ("conv_gather_mfma_glds (direct fwd / dgrad / stride-2 / transposed conv on the LDS-DMA kernel, Cin >= 32)", 1.0,
        "conv_gather_mfma_glds"),
    5: ("conv_gather_mfma (register-staged kernel of the Cin < 32 layers: condition-noise convs, 9-channel D input)", 1.0,
        "conv_gather_mfma"),
    1: ("conv_wgrad_mfma (direct weight gradient)", 1.0, "conv_wgrad_mfma"),
    2: ("wino_gemm_mfma (Winograd F(2x2,3x3) fwd / dgrad GEMM + fused output transform and epilogue)", WINOGRAD_EXECUTED,
        "wino_gemm_mfma"),
    3: ("conv_wgrad_mfma in planes mode (Winograd F(3x3,2x2) weight-gradient GEMM)", WINOGRAD_EXECUTED, "conv_wgrad_mfma"),
    6: ("conv_gather_mfma_glds<f16> + conv_halo_f16 (f16 fwd / dgrad / stride-2 / transposed conv, v_mfma_f32_32x32x16_f16; layers with "
        "<= 64 contraction and output channels on the halo kernel: input patch + halo staged in LDS once, taps by shifted LDS reads)", 1.0,
        "conv_gather_mfma_glds_f16"),
    7: ("conv_wgrad_mfma<f16> + conv_wgrad_halo_f16 (f16 weight gradient, fp32 accumulation; 256x256 tiles for multiples of 256 channels, "
        "persistent halo kernel with transposing LDS reads for <= 32 channels)", 1.0, "conv_wgrad_mfma_f16"),
    8: ("conv_gather_mfma_glds<float, bf16x3> (fp32 direct fwd / dgrad / stride-2 / transposed conv on the bf16 matrix cores: exact "
        "3-way bf16 split of both operands, 6 v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation)", BF16X3_EXECUTED,
        "conv_gather_mfma_glds_x3"),
    9: ("conv_wgrad_mfma<float, bf16x3> (fp32 weight gradient on the bf16 matrix cores, same split)", BF16X3_EXECUTED,
        "conv_wgrad_mfma_x3"),
    10: ("wino_gemm_x3 (Winograd F(2x2,3x3) fwd / dgrad GEMM on the bf16 matrix cores: bf16x3 split of V in the kernel, pre-split "
         "U, fused output transform and epilogue)", BF16X3_EXECUTED * WINOGRAD_EXECUTED, "wino_gemm_x3"),
    11: ("conv_wgrad_mfma<float, bf16x3> in planes mode (Winograd F(3x3,2x2) weight-gradient GEMMs on the bf16 matrix cores)",
         BF16X3_EXECUTED * WINOGRAD_EXECUTED, "conv_wgrad_mfma_x3"),
    13: ("conv_gather_mfma_glds<float, f16x2> (fp32 direct fwd / dgrad / stride-2 / transposed conv on the f16 matrix cores: two f16 terms per "
         "operand under per-row power-of-two scales — running row exponent with exact accumulator rescale, per-row weight exponents — 3 "
         "v_mfma_f32_32x32x16_f16 products per fp32 product, fp32 accumulation; the time includes the guarded bf16x3 twin launch, a no-op "
         "unless an operand left the precision window)", F16X2_EXECUTED, "conv_gather_mfma_glds_h2"),
    14: ("wino_gemm_h2 (Winograd F(2x2,3x3) fwd / dgrad GEMM on the f16 matrix cores: f16x2 split of V in the kernel under one running "
         "exponent per tile row, pre-split U, fused output transform and epilogue)", F16X2_EXECUTED * WINOGRAD_EXECUTED, "wino_gemm_h2"),
    15: ("conv_wgrad_mfma<float, f16x2> (fp32 weight gradient on the f16 matrix cores: both operands split in the kernel under running "
         "per-channel exponents, 3 products)", F16X2_EXECUTED, "conv_wgrad_mfma_h2"),
    16: ("conv_wgrad_mfma<float, f16x2> in planes mode (Winograd F(3x3,2x2) weight-gradient GEMMs on the f16 matrix cores)",
         F16X2_EXECUTED * WINOGRAD_EXECUTED, "conv_wgrad_mfma_h2"),
    17: ("conv_gather_mfma_glds<float, f16x2> in the tap-dense K order (3x3 layers with 8..28 contraction channels)", F16X2_EXECUTED, None),
    12: ("conv_gather_mfma_glds<float, bf16x3> in the tap-dense K order (3x3 layers with 8..28 contraction channels: the 6->12->24 "
         "condition-noise convs and the 24->C layers that inject their result; same kernels as the bf16x3 direct family, so no separate "
         "PMC traffic)", BF16X3_EXECUTED, None),
}
FAMILY_KEYS = {0: "roofline_conv_direct", 1: "roofline_wgrad_direct", 2: "roofline_conv_winograd", 3: "roofline_wgrad_winograd",
               5: "roofline_conv_direct_small_cin", 6: "roofline_conv_f16", 7: "roofline_wgrad_f16",
               8: "roofline_conv_direct_bf16x3", 9: "roofline_wgrad_direct_bf16x3", 10: "roofline_conv_winograd_bf16x3",
               11: "roofline_wgrad_winograd_bf16x3", 12: "roofline_conv_direct_bf16x3_tapdense", 13: "roofline_conv_direct_f16x2",
               14: "roofline_conv_winograd_f16x2", 15: "roofline_wgrad_direct_f16x2", 16: "roofline_wgrad_winograd_f16x2",
               17: "roofline_conv_direct_f16x2_tapdense"}
FAMILY_PEAK = {6: PEAK_F16_MFMA_TFLOPS, 7: PEAK_F16_MFMA_TFLOPS, 8: PEAK_F16_MFMA_TFLOPS, 9: PEAK_F16_MFMA_TFLOPS,
               10: PEAK_F16_MFMA_TFLOPS, 11: PEAK_F16_MFMA_TFLOPS, 12: PEAK_F16_MFMA_TFLOPS, 13: PEAK_F16_MFMA_TFLOPS,
               14: PEAK_F16_MFMA_TFLOPS, 15: PEAK_F16_MFMA_TFLOPS, 16: PEAK_F16_MFMA_TFLOPS,
               17: PEAK_F16_MFMA_TFLOPS}


def load_pmc():
    """HBM bytes per launch per kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot be read
    from inside the timed run).  The file names the commit and the command it was taken at; bench.py only relays it."""
    try:
        with open(PMC_JSON) as fh:
            return json.load(fh)
    except Exception:
        return None


def family_ceiling(fam, exec_frac, mfma_peak):
    """(TFLOP/s ceiling of the ALGORITHMIC rate, its name): the dense peak of the MFMA type the family feeds, divided by the MFMA
    FLOPs it executes per algorithmic (direct-convolution, fp32) FLOP."""
    if fam in (8, 9, 12):
        return mfma_peak / exec_frac, "bf16x3 fp32-exact = 2500 / 6 (six bf16 MFMA products per fp32 product, bf16 dense peak 2500 TFLOP/s)"
    if fam in (13, 15, 17):
        return mfma_peak / exec_frac, "f16x2 = 2500 / 3 (three f16 MFMA products per fp32 product, f16 dense peak 2500 TFLOP/s)"
    if fam in (14, 16):
        return mfma_peak / exec_frac, "Winograd on f16x2 = 2500 / (3 * 16/36) (F(2x2,3x3) executes 16/36 of the products, each as three f16 MFMA products)"
    if fam in (10, 11):
        return mfma_peak / exec_frac, "Winograd on bf16x3 = 2500 / (6 * 16/36) (F(2x2,3x3) executes 16/36 of the products, each as six bf16 MFMA products)"
    if fam in (2, 3):
        return mfma_peak / exec_frac, "Winograd on fp32 MFMA = 157.3 / (16/36)"
    if fam in (6, 7):
        return mfma_peak, "f16 MFMA dense peak 2500 TFLOP/s"
    return mfma_peak, "fp32-input MFMA dense peak 157.3 TFLOP/s"


def roofline_objects(ops, steps, wall_s):
    """Per-family HIP-event timings -> the `roofline` object of the dominant kernel family + one object per other MFMA family.

    SURVEY §8(d): `achieved` = ALGORITHMIC FLOPs (direct-convolution count, fp32) per launch / average launch duration; `peak` =
    the named ceiling of that algorithmic rate (`ceiling`: the dense MFMA peak of the type the kernels feed divided by the MFMA
    FLOPs executed per algorithmic FLOP); `frac` = achieved / peak.  The raw matrix-pipe view is next to it: `executed_achieved`
    (MFMA FLOP/s actually issued), `executed_peak` (data-sheet dense peak of that MFMA type), `executed_frac` — numerically the
    same ratio — and `frac_of_mfma_type_peak` = algorithmic rate / that data-sheet peak (what the fp32 workload gets out of a bf16
    pipe that is 16x faster than the fp32 one)."""
    pmc = load_pmc()
    digest = kernel_source_digest()
    objs, executed_flops, executed_peak_s, mfma_ms = {}, 0.0, 0.0, 0.0
    for fam, (name, exec_frac, pmc_key) in FAMILIES.items():
        ms, fl, n = ops.prof_read(fam)
        if n == 0 or ms <= 0:
            continue
        mfma_peak = FAMILY_PEAK.get(fam, PEAK_F32_MFMA_TFLOPS)
        executed_flops += fl * exec_frac
        executed_peak_s += fl * exec_frac / (mfma_peak * 1e12)  # seconds this work takes at the peak of the MFMA type it ran on
        mfma_ms += ms
        alg = fl / (ms * 1e-3) / 1e12
        ceiling, ceiling_name = family_ceiling(fam, exec_frac, mfma_peak)
        o = {"bound": "mfma", "kernel": name, "achieved": alg, "peak": ceiling, "unit": "TFLOP/s", "frac": alg / ceiling,
             "ceiling": ceiling_name, "executed_achieved": alg * exec_frac, "executed_peak": mfma_peak,
             "executed_frac": alg * exec_frac / mfma_peak, "frac_of_mfma_type_peak": alg / mfma_peak,
             "executed_mfma_flop_per_algorithmic_flop": exec_frac, "traffic": None, "launches": n, "avg_ms": ms / n,
             "gpu_ms_per_step": ms / steps, "algorithmic_flop_per_launch": fl / n}
        fam_pmc = (pmc or {}).get("families", {}).get(pmc_key)
        if fam_pmc:
            o["traffic"] = fam_pmc["hbm_bytes_per_launch"]
            stale = pmc.get("kernel_source_digest") != digest
            o["traffic_source"] = {"file": "profiles/pmc_traffic.json", "kernels": fam_pmc.get("kernels"),
                                   "launches_profiled": fam_pmc.get("launches"), "commit": pmc.get("commit"),
                                   "kernel_source_digest": pmc.get("kernel_source_digest"), "this_run_digest": digest,
                                   "stale": stale, "command": pmc.get("command"),
                                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (2*FETCH + WRITE, gfx950 correction), "
                                           "family average over one profiled training iteration"
                                           + ("; STALE: the kernel sources changed since it was measured" if stale else "")}
            # the same launches against the OTHER roof: measured HBM bytes per average launch over its average duration.  fp32 tensors make these
            # kernels memory-heavy: for the f16x2 families the two floors (time at the MFMA ceiling, time at 8 TB/s) lie within ~10 % of each other
            hbm_gbs = o["traffic"] / (o["avg_ms"] * 1e-3) / 1e9
            o["hbm_view"] = {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                             "floor_ms_at_hbm_peak": o["traffic"] / (PEAK_HBM_GBS * 1e9) * 1e3,
                             "floor_ms_at_mfma_ceiling": o["algorithmic_flop_per_launch"] / (ceiling * 1e12) * 1e3,
                             "note": "measured (PMC) HBM bytes of the family's average launch / its average duration; floors per average launch"}
        objs[fam] = o
    out = {}
    if objs:
        dom = max(objs, key=lambda f: objs[f]["gpu_ms_per_step"])
        out["roofline"] = objs.pop(dom)
        for fam, o in objs.items():
            out[FAMILY_KEYS[fam]] = o
    ms, by, n = ops.prof_read(4)
    if n:
        o = {"bound": "hbm", "kernel": "wino_input_transform / wino_gy_transform", "achieved": by / (ms * 1e-3) / 1e9,
             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None, "launches": n,
             "avg_ms": ms / n, "gpu_ms_per_step": ms / steps}
        fam_pmc = (pmc or {}).get("families", {}).get("wino_transforms")
        if fam_pmc:
            o["traffic"] = fam_pmc["hbm_bytes_per_launch"]
            o["traffic_stale"] = pmc.get("kernel_source_digest") != digest
        out["roofline_winograd_transforms"] = o
    # what the MFMA pipe really did over the WALL time of the timed region (Winograd's skipped multiplies not counted)
    out["executed_mfma_frac_wall"] = {
        "executed_tflop_per_step": executed_flops / steps / 1e12,
        "achieved": executed_flops / wall_s / 1e12, "unit": "TFLOP/s",
        "frac": executed_peak_s / wall_s,
        "mfma_kernel_ms_per_step": mfma_ms / steps,
        "frac_while_mfma_kernels_run": executed_peak_s / (mfma_ms * 1e-3) if mfma_ms else None,
        "note": "MFMA FLOPs actually executed (direct kernels: all; Winograd GEMMs: 16/36 of the algorithmic count), each priced "
                "at the dense peak of the MFMA type it ran on (fp32 157.3 TF, f16 / bf16 2500 TF; bf16x3: 6 executed per algorithmic, f16x2: 3): time at peak / wall time of the "
                "timed region"}
    return out


def rasterize_roofline(batch, dev):
    """`roofline_rasterize`: gif_rasterize_f32 (gif_amd/csrc/rasterize.hip) on the reference's own test mesh (body.obj, one random
    yaw per image) at 256x256 — SURVEY §8(d): triangles/s, covered pixels/s and the algorithmic bytes (36 F + 20 H W) B per
    call against HBM peak — timed OUTSIDE the training-step region (HIP-graph replays, so that the figure is GPU time and not
    Python launch overhead), with the C oracle (oracle/rasterize_ref.c, ONE host thread) on the same meshes as its CPU baseline."""
    from tools import raster_bench as rb
    res = 256
    v, f = rb.body_mesh(batch)
    fv_np = rb.face_vertices_np(v, f, res)
    fv = torch.from_numpy(fv_np).to(dev)
    ms, out = rb.time_hip(fv, res)
    F_ = f.shape[1]
    covered = int((out[1] >= 0).sum())
    alg_bytes = (36.0 * F_ + 20.0 * res * res) * batch
    o = {"bound": "hbm", "kernel": "raster_bin + raster_tiles (LDS-resident 64x64-pixel z-buffer tiles)", "achieved": alg_bytes / ms / 1e6,
         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg_bytes / ms / 1e6 / PEAK_HBM_GBS, "traffic": None, "avg_ms": ms,
         "workload": f"body.obj (F = {F_} faces) x {batch} images at {res}x{res}", "Mtri_per_s": batch * F_ / ms / 1e3,
         "covered_Mpix_per_s": covered / ms / 1e3, "covered_frac": covered / (batch * res * res),
         "algorithmic_bytes_per_call": alg_bytes,
         "note": "latency / launch bound at this size (a call is ~16 MB of faces + 8 MB of depth): frac of HBM peak is reported as "
                 "SURVEY §8(d) asks, the absolute time is what matters (bit-exact vs the C oracle: tests/test_gpu_kernels.py)"}
    try:
        s_img, n_img, _ = rb.time_oracle(fv_np, res)
        o["cpu_baseline"] = {"value": F_ / s_img / 1e6, "unit": "Mtri/s", "cores": 1, "kind": "port",
                             "sample": f"oracle/rasterize_ref.c on the first {n_img} images of the same batch: {s_img * 1e3:.2f} ms per image",
                             "gpu_over_one_cpu_thread": (s_img * 1e3) / (ms / batch)}
    except Exception as e:  # the oracle .so is test infrastructure: never block the GPU numbers
        o["cpu_baseline"] = {"value": None, "unit": "Mtri/s", "cores": 1, "kind": "port", "sample": f"oracle not available ({type(e).__name__})"}
    return o


class MeshConditions:
    """Config 3: a posed synthetic mesh of FLAME size rendered to the 6-channel condition with the HIP vertex-normal and
    rasteriser kernels (gif_amd.render.render_condition) — called INSIDE the timed region."""

    def __init__(self, batch, res, dev, seed):
        import numpy as np
        from gif_amd import render
        self.render, self.res = render, res
        m = np.load(os.path.join(ROOT, "tests", "golden", "body_mesh.npz"))
        rng = np.random.RandomState(seed)
        verts = []
        for _ in range(batch):
            a = rng.uniform(-0.4, 0.4)
            rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
            verts.append((m["vertices"] @ rot.T).astype(np.float32))
        self.v = torch.from_numpy(np.stack(verts)).to(dev)
        self.f = torch.from_numpy(m["faces"]).to(dev)
        self.cam = torch.tensor([[0.95, 0.0, 0.35]] * batch, device=dev)
        self.tex = (self.v - self.v.amin(dim=1, keepdim=True)) / (self.v.amax(dim=1, keepdim=True) - self.v.amin(dim=1, keepdim=True))

    def __call__(self):
        v_ndc = self.render.batch_orth_proj(self.v, self.cam)
        return self.render.render_condition(v_ndc, self.f, self.tex, self.res, self.res)


def texture_interp_loss(batch, res, dev):
    """--texture-interp: losses.InterpolatedTextureLoss with its FLAME-dependent parts injected: the condition
    of the interpolated FLAME batch is RENDERED inside the timed region (vertex normals + two rasteriser passes), the generator runs
    on batch - 1 images of one fixed identity, their textures are lifted into the 256x256 UV space and batch - 1 random pairs are
    compared (loss_functions/losses.py:162-243)."""
    import numpy as np
    from gif_amd import data, losses, render
    from gif_amd.texture_space import FlameTextureSpace
    m = np.load(os.path.join(ROOT, "tests", "golden", "body_mesh.npz"))
    flame = data.SyntheticFlame(m["vertices"], dev, seed=5)
    faces = torch.from_numpy(m["faces"]).to(dev)
    v = torch.from_numpy(m["vertices"]).to(dev)
    vtx_tex = (v - v.amin(dim=0, keepdim=True)) / (v.amax(dim=0, keepdim=True) - v.amin(dim=0, keepdim=True))
    tex_dec = FlameTextureSpace(data.synthetic_texture_data(m["faces"], fill=0.6, seed=6), None, flame=flame, faces=faces).to(dev)
    ys, xs = np.meshgrid(np.linspace(-1, 1, 256), np.linspace(-1, 1, 256), indexing="ij")
    face_mask = torch.from_numpy((((xs / 0.8) ** 2 + (ys / 0.9) ** 2) <= 1).astype(np.float32))[None, None].to(dev)
    return losses.InterpolatedTextureLoss(batch, face_mask, flm_tex_dec=tex_dec,
                                          render_condition=render.FlameConditionRenderer(flame, faces, vtx_tex, res, res))


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_command(n, argv, port):
    """The command line `python bench.py --gpus N ...` re-executes itself as when no launcher set WORLD_SIZE: N ranks of ONE node
    under torch.distributed.run (the form the task contract names), rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (train.py:344-358 wraps the networks in DataParallel inside
    ONE process; here every GPU gets its own process): spawn the N ranks and relay rank 0's JSON line.  Fails before spawning
    anything when the node has fewer than N GPUs."""
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        sys.exit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible on this node (one rank per GPU; "
                 f"HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')})")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launcher_command(n, argv, free_port())
    print("bench.py: self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    rc = subprocess.run(cmd, env=env).returncode
    if rc:
        sys.exit(rc)


def main():
    args = parse()
    if args.cpu_baseline_worker:
        r, st, b, th = (int(v) for v in args.cpu_baseline_worker.split(","))
        return cpu_baseline_worker(r, st, b, th)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus, sys.argv[1:])  # `python bench.py --gpus N`: one rank per GPU under torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs MI355X GPUs (no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("GIF_FORCE_DIST") == "1"  # (forced: RCCL code paths on a 1-GPU box, tests)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    from gif_amd import ops
    if args.fp32_mfma:
        ops.set_fp32_mfma_mode(args.fp32_mfma)
    from gif_amd.discriminator import Discriminator
    from gif_amd.generator import StyledGenerator
    from gif_amd.train_step import GifTrainer, flops_per_image

    res_step = {64: 4, 128: 5, 256: 6, 512: 7, 1024: 8}[args.res]
    torch.manual_seed(1234 + rank)  # per-rank seeds on purpose: GifTrainer broadcasts rank 0's state at construction
    with contextlib.redirect_stdout(io.StringIO()):
        kw = dict(embedding_vocab_size=args.vocab, rendered_flame_ascondition=True, normal_maps_as_cond=True,
                  core_tensor_res=4, n_mlp=8)
        G = StyledGenerator(**kw)
        G_ema = StyledGenerator(**kw)
        D = Discriminator(size=args.res, num_color_chnls=9, channel_multiplier=2)
    G_ema.load_state_dict(G.state_dict())
    G, G_ema, D = G.to(dev), G_ema.to(dev), D.to(dev)
    tex_loss = texture_interp_loss(args.batch, args.res, dev) if args.texture_interp else None
    trainer = GifTrainer(G, D, G_ema, step=res_step, alpha=1.0, r1_every=args.r1_every, gen_reg_type=args.gen_reg,
                         texture_loss=tex_loss, max_ids=args.vocab,
                         act_dtype=torch.float16 if args.dtype == "f16" else None,
                         overlap_comm=False if args.no_overlap_comm else None,
                         reuse_generator_forward=args.reuse_generator_forward,
                         fuse_d_passes=False if args.two_call_d else None)

    from gif_amd.data import SyntheticBatches
    B = args.batch
    batches = SyntheticBatches(B, args.res, args.vocab, dev, seed=1234, rank=rank)  # every rank draws its own data
    mesh = MeshConditions(B, args.res, dev, seed=99 + rank) if args.render_cond else None

    flame_gen = torch.Generator(device=dev).manual_seed(4321 + rank)

    def batch():
        b = next(batches)
        if tex_loss is not None:  # FLAME labels of the batch (dataset_loaders.py: flm_lbls), synthetic
            from gif_amd.data import synthetic_flame_labels
            lbl = synthetic_flame_labels(B, dev, flame_gen)
            lbl[:, 157:159] = 0.0  # (the camera offset of the labels is made for a head mesh; the stand-in template is centred)
            b = b + (lbl,)
        return b

    def run_step(it, b):
        real, cond, idx = b[:3]
        if mesh is not None:
            cond = mesh()  # config 3: rasterised condition, inside the timed region
        return trainer.step(it, real, cond, idx, flame_batch=b[3] if len(b) > 3 else None)

    # R1 share of the timed region = steps / r1_every (review item): r1_start_iteration
    it = r1_start_iteration(args.steps, args.warmup, args.r1_every)
    it0 = it
    for _ in range(args.warmup):
        run_step(it, batch())
        it += 1
    r1_timed = sum(1 for k in range(args.steps) if args.r1_every and (it + k + 1) % args.r1_every == 0)
    r1_warm = sum(1 for k in range(args.warmup) if args.r1_every and (it0 + k + 1) % args.r1_every == 0)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for bk in (trainer.g_bucket, trainer.d_bucket):
        bk.comm_events = []  # (start, end) event pairs around every wait for an exchange: exposed communication time
    data = [batch() for _ in range(min(args.steps, 4))]  # synthetic batches resident in HBM before the timed region
    prof_steps = 0
    if not args.no_prof:
        for fam in range(18):
            ops.prof_read(fam)
    h2_mode = args.dtype != "f16" and ops.get_fp32_mfma_mode() == "f16x2"
    if h2_mode:
        ops.h2_fallback_stats(reset=True)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sampled = not args.no_prof and k % max(args.prof_every, 1) == 0
        if sampled:  # instrument this step's launches (the events are read after the timed region)
            ops.prof_enable(True)
            prof_steps += 1
        run_step(it, data[k % len(data)])
        if sampled:
            ops.prof_enable(False)
        it += 1
    trainer.flush()
    sync()
    dt = time.perf_counter() - t0

    comm_ms = sum(e0.elapsed_time(e1) for bk in (trainer.g_bucket, trainer.d_bucket) for e0, e1 in bk.comm_events)
    t = torch.tensor([dt, comm_ms], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, comm_ms = t[0].item(), t[1].item()

    replicas = {"ranks": world, "backend": dist.get_backend() if use_dist else None}
    if args.check_replicas or world > 1:  # (always with > 1 rank: the digests are what shows that the exchange kept the replicas equal)
        import hashlib
        h = hashlib.sha1()
        for m in (G, D, G_ema):
            for p in m.parameters():
                h.update(p.detach().cpu().numpy().tobytes())
        mine = h.hexdigest()
        allr = [mine]
        if use_dist:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
        replicas.update({"param_digest": allr[0], "replicas_identical": len(set(allr)) == 1, "ranks": len(allr)})

    if rank == 0:
        imgs = world * B * args.steps
        value = imgs / dt
        fl_img = flops_per_image(args.res, args.r1_every, generator_forwards=1 if args.reuse_generator_forward else 2,
                                 extra_generator_fwd_bwd=(B - 1) / B if args.texture_interp else 0.0)
        step_tflops = value * fl_img / 1e12 / world
        f16 = args.dtype == "f16"
        fp32_mode = ops.get_fp32_mfma_mode()
        # peak of the MFMA type the contractions are actually fed to: the f16 / bf16 pipe unless --fp32-mfma native
        peak = PEAK_F32_MFMA_TFLOPS if (not f16 and fp32_mode == "native") else PEAK_F16_MFMA_TFLOPS
        workload = (f"GIF run-29 G+D training iteration, {args.res}x{args.res}, batch {B}/GPU, R1 every {args.r1_every}th step, "
                    + ("f16 activations / f16 MFMA with fp32 accumulation, fp32 weights + demodulation, dynamic loss scaling "
                       "(BASELINE configs[4])" if f16 else
                       "fp32 tensors and fp32 accumulation; contractions on "
                       + ("the f16 matrix cores via a two-term f16 split under per-row power-of-two scales (f16x2: 3 products per fp32 product, "
                          "error vs fp64 <= the native fp32 MFMA path on in-window operands, guarded bf16x3 fallback otherwise) for the direct "
                          "fwd / dgrad kernels, the weight gradients and the Winograd GEMMs with >= 24 contraction channels; bf16x3 (6 products) "
                          "in a tap-dense K order for the 3x3 convs with 8..28 contraction channels; the 9-channel D input layer "
                          "(1x1), ToRGB's data gradient and the small-channel weight gradients on native fp32 MFMA"
                          if fp32_mode == "f16x2" else
                          "the bf16 matrix cores via the exact 3-way bf16 split (bf16x3: 6 products per fp32 product, error vs fp64 <= "
                          "the native fp32 MFMA path) for every direct, weight-gradient and Winograd GEMM with >= 24 contraction "
                          "channels and, in a tap-dense K order, the 3x3 convs with 8..28 (the 6->12->24 condition-noise convs); the "
                          "9-channel D input layer (1x1), ToRGB's data gradient and the small-channel weight gradients on native fp32 MFMA"
                          if fp32_mode == "bf16x3" else "native fp32 MFMA") + " (BASELINE configs[1]/[3] shape)"))
        if args.reuse_generator_forward:
            workload += ("; NOT the reference's call order: ONE generator forward per iteration shared by the D and G steps "
                         "(identical results, one forward less of work)")
        if args.render_cond:
            workload += "; condition rasterised from a posed mesh inside the timed region (configs[2])"
        if args.gen_reg.upper() != "NONE":
            workload += f"; generator regulariser {args.gen_reg.upper()}"
        if args.texture_interp:
            workload += (f"; texture-space interpolation loss on every generator step (train.py:222-238: {B - 1} extra generator images "
                         "from a condition rendered in the timed region, UV texture lifting, pairwise loss; synthetic FLAME stand-in)")
        out = {
            "metric": f"G+D train-step images/sec at {args.res}x{args.res}, batch {B}/GPU",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * B, "resolution": args.res, "parallelism": f"dp{world}",
                       "fp32_mfma": None if f16 else fp32_mode,
                       "non_default_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("GIF_")},
                       "f16x2_fallback_launches_timed": ops.h2_fallback_stats() if h2_mode else None,
                       "r1_iterations_timed": r1_timed, "r1_iterations_warmup": r1_warm, "first_iteration_index": it0,
                       "algorithmic_tflop_per_image": fl_img / 1e12,
                       "grad_bucket_mb": {"G": trainer.g_bucket.flat.numel() * 4 / 1e6, "D": trainer.d_bucket.flat.numel() * 4 / 1e6},
                       "overlap_comm": bool(trainer.overlap_comm),
                       "d_step_discriminator_passes": ("one pass over [real; fake] (minibatch-stddev per half; R1 iterations: two calls)"
                                                       if trainer.fuse_d_passes else "two calls (train.py:142, :169)"),
                       "peak_hbm_allocated_gb": torch.cuda.max_memory_allocated() / 2 ** 30},
            "comm_exposed_ms": comm_ms / args.steps,
            "comm_note": ("one RCCL all-reduce (AVG) per optimiser step over each flat gradient bucket, enqueued right after the backward and "
                          "waited for where the network is next used; comm_exposed_ms = GPU time per step the compute stream spent in those "
                          "waits (HIP events around FlatGradBucket.wait, max over ranks)" if use_dist else
                          "single process: no gradient exchange"),
            "step_mfma_roofline": {"achieved": step_tflops, "peak": peak, "unit": "TFLOP/s",
                                   "frac": step_tflops / peak,
                                   "frac_of_fp32_input_mfma_peak": step_tflops / PEAK_F32_MFMA_TFLOPS,
                                   "note": "whole step, ALGORITHMIC direct-convolution FLOPs (Winograd executes fewer, f16x2 three / bf16x3 six "
                                           "times more: see executed_mfma_frac_wall) incl. HBM-bound kernels, optimiser, host; per GPU.  "
                                           "peak = the dense peak of the MFMA type the contractions are fed to (f16 / bf16: 2500; --fp32-mfma "
                                           "native: 157.3).  frac_of_fp32_input_mfma_peak is context only (the fp32-input pipe is 16x slower "
                                           "and is not what these kernels run on; it can exceed 1)"},
        }
        if not args.no_prof:
            out.update(roofline_objects(ops, prof_steps, dt * prof_steps / args.steps))
            out["roofline_sampling"] = {"instrumented_steps": prof_steps, "of": args.steps,
                                        "note": f"HIP events around every MFMA / Winograd-transform launch of every {max(args.prof_every, 1)}th timed step; "
                                                "per-step figures are per instrumented step, executed_mfma_frac_wall uses the mean step time"}
            try:
                out["roofline_rasterize"] = rasterize_roofline(B, dev)
            except Exception as e:  # never lose the headline line to the side measurement
                out["roofline_rasterize"] = {"error": f"{type(e).__name__}: {e}"}
        out["replicas"] = replicas
        if f16:
            out["loss_scaler"] = {"g_scale": trainer.g_scaler.scale.item(), "d_scale": trainer.d_scaler.scale.item(),
                                  "skipped_g_steps": trainer.g_scaler.skipped.item(), "skipped_d_steps": trainer.d_scaler.skipped.item()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.res, res_step, args.cpu_batch, args.cpu_threads, args.cpu_timeout)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()  # rank 0 does its side measurements above: the other ranks leave the process group together with it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
