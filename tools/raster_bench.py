"""Rasteriser throughput on one MI355X (SURVEY §8(d) "Algorithmic bytes": (36 F + 20 H W) B bytes per launch): triangles/s,
covered pixels/s and GB/s of gif_rasterize_f32, next to
  * the round-2 kernel (one lane per face over its whole bounding box; tools/probes/rasterize_v1.hip, built here as
    tools/probes/libraster_v1.so) — results must be bit-identical, and
  * the C oracle (oracle/rasterize_ref.c, one host thread) on the same meshes = the CPU baseline of this kernel.
Usage (GPU): python tools/raster_bench.py [--batch 32] [--json out.json]
bench.py imports workloads() / time_hip() / time_oracle() for its `roofline_rasterize` object.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
V1_SRC = os.path.join(ROOT, "tools", "probes", "rasterize_v1.hip")
V1_LIB = os.path.join(ROOT, "tools", "probes", "libraster_v1.so")
PEAK_HBM_GBS = 8000.0


def build_v1():
    if os.path.exists(V1_LIB) and os.path.getmtime(V1_LIB) >= os.path.getmtime(V1_SRC):
        return
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", f"-I{ROOT}/gif_amd/csrc",
                           f"-I{ROOT}/include", V1_SRC, "-o", V1_LIB])


def body_mesh(batch, seed=0):
    """The reference's own test mesh (my_utils/standard_rasterize_cuda/data/obj/body.obj via tests/golden/body_mesh.npz,
    demo_vert_visibility.py:12-22: verts * 0.8), one random yaw per sample."""
    m = np.load(os.path.join(ROOT, "tests", "golden", "body_mesh.npz"))
    rng = np.random.RandomState(seed)
    vs = []
    for _ in range(batch):
        a = rng.uniform(-0.5, 0.5)
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        vs.append((m["vertices"].astype(np.float32) * np.float32(0.8)) @ rot.T)
    return np.stack(vs).astype(np.float32), np.repeat(m["faces"][None].astype(np.int32), batch, 0)


def with_backdrop(v, f, n_big=4):
    """The same meshes in front of `n_big` screen-filling triangles (two quads of the far plane): the case that serialises a
    lane-per-face rasteriser."""
    B, V = v.shape[:2]
    zfar = v[..., 2].max() + 0.5
    quad = np.array([[-1.2, -1.2, zfar], [-1.2, 1.2, zfar], [1.2, -1.2, zfar], [1.2, 1.2, zfar]], np.float32)
    extra_v, extra_f = [], []
    for k in range(n_big // 2):
        q = quad.copy()
        q[:, 2] += 0.1 * k
        base = V + 4 * k
        extra_v.append(q)
        # both windings of each half so that one of them is front-facing whatever the convention
        extra_f += [[base, base + 1, base + 2], [base + 2, base + 1, base + 3], [base, base + 2, base + 1], [base + 2, base + 3, base + 1]]
    ev = np.concatenate(extra_v)[None].repeat(B, 0)
    ef = np.array(extra_f, np.int32)[None].repeat(B, 0)
    return np.concatenate([v, ev], 1), np.concatenate([f, ef], 1)


def random_medium(batch, nfaces=3000, size=0.25, seed=1):
    """Random triangles of ~size (NDC units): bounding boxes of a few hundred to a few thousand pixels at 256^2."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-0.9, 0.9, (batch, nfaces, 1, 3)).astype(np.float32)
    v = (c + rng.uniform(-size, size, (batch, nfaces, 3, 3)).astype(np.float32)).reshape(batch, nfaces * 3, 3)
    v[..., 2] = rng.uniform(0.0, 1.0, v.shape[:2]).astype(np.float32)
    f = np.arange(nfaces * 3, dtype=np.int32).reshape(1, nfaces, 3).repeat(batch, 0)
    return v, f


def workloads(batch):
    bv, bf = body_mesh(batch)
    return [("body.obj", bv, bf, 256), ("body.obj", bv, bf, 512), ("body.obj + 4 screen-filling faces",) + with_backdrop(bv, bf) + (256,),
            ("3000 random medium faces",) + random_medium(batch) + (256,)]


def face_vertices_np(v, f, res):
    """[B,F,3,3] float32 in pixel units, built with the oracle's helpers so that the HIP kernels and the C oracle see the same bits."""
    from oracle import rasterize_oracle as ro
    return ro.face_vertices(ro.to_image_space(v, res, res), f)


def _timeit(fn, calls=20, replays=5):
    """GPU time per call: `calls` calls captured into one HIP graph (a call is two ~5 us launches: eager enqueueing from
    Python would measure the host), replayed `replays` times between two events."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()  # (per-stream state of the callee — the rasteriser's persistent workspace — exists before the capture starts)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(calls):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (calls * replays)


def time_hip(fv, res, lib=None):
    """(ms per rasteriser call, (depth, tri, bary)) — buffers pre-allocated, re-initialised by a
    device copy outside the timed call like a caller's new_buffers()."""
    from gif_amd import ops
    from gif_amd import standard_rasterize as sr
    B, F = fv.shape[:2]
    d, t, b = sr.new_buffers(B, res, res, "cuda")
    if lib is None:
        def call():
            ops.rasterize(fv, d, t, b, res, res)
    else:
        ws = torch.empty((lib.v1_gif_rasterize_workspace_bytes(B, res, res) // 8,), device="cuda", dtype=torch.int64)

        def call():
            rc = lib.v1_gif_rasterize_f32(fv.data_ptr(), d.data_ptr(), t.data_ptr(), b.data_ptr(), B, F, res, res, ws.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    call()
    out = (d.clone(), t.clone(), b.clone())

    # chained calls are idempotent (the keys are re-seeded from the depth buffer): no re-initialisation between timed calls
    return _timeit(call), out


def time_oracle(fv_np, res, max_images=4):
    """Single-thread C oracle on the first images of the batch: (seconds per image, images timed, its buffers)."""
    from oracle import rasterize_oracle as ro
    n = min(max_images, fv_np.shape[0])
    fv = np.ascontiguousarray(fv_np[:n])
    d, t, b = ro.new_buffers(n, res, res)
    t0 = time.perf_counter()
    ro.standard_rasterize(fv, d, t, b, res, res)
    return (time.perf_counter() - t0) / n, n, (d, t, b)


def load_v1():
    build_v1()
    lib = ctypes.CDLL(V1_LIB)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.v1_gif_rasterize_workspace_bytes.restype = ctypes.c_int64
    lib.v1_gif_rasterize_workspace_bytes.argtypes = [I, I, I]
    lib.v1_gif_rasterize_f32.restype = I
    lib.v1_gif_rasterize_f32.argtypes = [P, P, P, P, I, I, I, I, P, P]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--json", type=str, default=None)
    a = ap.parse_args()
    v1 = load_v1()
    rows = []
    for name, v, f, res in workloads(a.batch):
        B, F = f.shape[:2]
        fv_np = face_vertices_np(v, f, res)
        fv = torch.from_numpy(fv_np).cuda()
        ms2, o2 = time_hip(fv, res)
        ms1, o1 = time_hip(fv, res, v1)
        same = all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(o1, o2))
        s_img, n_or, o_or = time_oracle(fv_np, res)
        same_or = all(np.array_equal(x[:n_or].cpu().numpy().view(np.int32), np.asarray(y).view(np.int32)) for x, y in zip(o2, o_or))
        covered = int((o2[1] >= 0).sum())
        alg_bytes = (36.0 * F + 20.0 * res * res) * B
        row = {"mesh": name, "res": res, "batch": B, "faces": F, "ms": ms2, "ms_round2_kernel": ms1, "speedup": ms1 / ms2,
               "Mtri_per_s": B * F / ms2 / 1e3, "covered_Mpix_per_s": covered / ms2 / 1e3, "Mpix_per_s": B * res * res / ms2 / 1e3,
               "algorithmic_GB_per_s": alg_bytes / ms2 / 1e6, "hbm_frac": alg_bytes / ms2 / 1e6 / PEAK_HBM_GBS,
               "covered_frac": covered / (B * res * res), "bit_identical_to_round2_kernel": same, "bit_identical_to_c_oracle": same_or,
               "cpu_oracle_ms_per_image_1_thread": s_img * 1e3, "cpu_oracle_Mtri_per_s": F / s_img / 1e6,
               "gpu_over_cpu_thread": (s_img * 1e3) / (ms2 / B)}
        rows.append(row)
        print(f"{name:36s} {res}^2 x{B} F={F}: {ms2:7.3f} ms (round-2 kernel {ms1:7.3f} ms, x{ms1 / ms2:.2f}) {row['Mtri_per_s']:8.0f} Mtri/s "
              f"{row['covered_Mpix_per_s']:7.0f} covered Mpix/s {row['algorithmic_GB_per_s']:7.1f} GB/s alg. | C oracle {s_img * 1e3:7.2f} ms/image "
              f"| identical: v1 {same} oracle {same_or}", flush=True)
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
