"""Raw (non-autograd) launches of the gfx950 kernels on torch device tensors.

Conventions: activations are torch tensors of LOGICAL shape [B,C,H,W] whose memory is NHWC
(torch.channels_last); fp32 (the reference dtype) with C % 4 == 0, or float16 (BASELINE config 5: f16 activations,
fp32 everything else) with C % 8 == 0 — the kernel family is chosen by the activation's dtype.  Weights, biases, per-sample
scales, FIR taps and every reduction result are fp32 in both cases.  Everything runs on torch's current HIP stream.
torch is used for device memory and streams only — all arithmetic is in libgif_hip.so.
"""
import ctypes
import functools
import os
import weakref
from typing import NamedTuple, Optional

import torch

from . import _lib

CL = torch.channels_last


def _stream():
    # current stream of the CURRENT device; _device_guard makes that the operands' device
    return torch.cuda.current_stream().cuda_stream


def _device_guard(fn):
    """Run `fn` with the operands' device current (launch + stream + per-device kernel state all follow it) and refuse
    operands that span devices.  One process per GPU never takes the slow branch."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        dev = None
        for a in args + tuple(kw.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise _lib.GifHipError(f"{fn.__name__}: operands on different devices ({dev} and {a.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapped


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def cpad(c: int, dtype=torch.float32) -> int:
    """Channel count of an activation holding c logical channels: multiple of 4 (fp32: 16-byte lanes) or 8 (f16)."""
    q = 8 if dtype == torch.float16 else 4
    return (c + q - 1) // q * q


def _sfx(dtype) -> str:
    return "_f16" if dtype == torch.float16 else "_f32"


def _fn(name: str, dtype):
    """C entry point of the kernel family matching the activation dtype (gif_<name>_f32 / gif_<name>_f16)."""
    return getattr(_lib.load(), "gif_" + name + _sfx(dtype))


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """fp32 or f16, device-resident, NHWC memory.  Fails loudly on CPU tensors: there is no CPU path."""
    if not x.is_cuda:
        raise _lib.GifHipError("gif_amd kernels need device tensors (no CPU fallback); got a CPU tensor")
    if x.dtype not in (torch.float32, torch.float16):
        raise _lib.GifHipError(f"gif_amd kernels take fp32 or f16 activations; got {x.dtype}")
    if x.dim() != 4:
        raise _lib.GifHipError(f"expected a 4-D activation, got shape {tuple(x.shape)}")
    if x.dtype == torch.float16 and x.shape[1] % 8:
        raise _lib.GifHipError(f"f16 activations need a channel count that is a multiple of 8, got {x.shape[1]}")
    return x.contiguous(memory_format=CL)


def _same_dtype(a, b, what):
    if a.dtype != b.dtype:
        raise _lib.GifHipError(f"{what}: operands of different dtypes ({a.dtype} and {b.dtype})")


def empty_nhwc(B, C, H, W, device, dtype=torch.float32):
    return torch.empty((B, C, H, W), device=device, dtype=dtype, memory_format=CL)


class ConvSpec(NamedTuple):
    """The underlying FORWARD convolution (see include/gif_hip.h)."""
    KH: int
    KW: int
    stride: int
    pad: int

    def small_hw(self, Hb, Wb):
        return ((Hb + 2 * self.pad - self.KH) // self.stride + 1, (Wb + 2 * self.pad - self.KW) // self.stride + 1)

    def big_hw(self, Hs, Ws):
        return ((Hs - 1) * self.stride + self.KH - 2 * self.pad, (Ws - 1) * self.stride + self.KW - 2 * self.pad)


def _geom(B, Hb, Wb, Cb, Hs, Ws, Cs, spec: ConvSpec):
    return _lib.ConvGeom(B, Hb, Wb, Cb, Hs, Ws, Cs, spec.KH, spec.KW, spec.stride, spec.pad)


# Gradient-producer fusions (gif_conv_epilogue ABI 2).  GIF_FUSE_GRAD=0 keeps every backward on the stand-alone passes (A/B).
FUSE_GRAD = _lib.knob("GIF_FUSE_GRAD", "1") != "0"


class GradFuse:
    """Work a gradient-producing launch does in its epilogue instead of leaving it to stand-alone passes (include/gif_hip.h,
    gif_conv_epilogue ABI 2):
      mask_src (+ slope, gain): the op's output is the gradient w.r.t. a tensor that a fused leaky ReLU produced — multiply by
                                gain * (mask_src > 0 ? 1 : slope), i.e. FusedLeakyReLU's backward (stylegan2_common_layers.py:22-39);
      want_colsum             : .colsum [C] = sum over all pixels of the stored output (the bias gradient of that layer);
      dot_src                 : .dot [B,C] = sum_hw contraction * dot_src, taken BEFORE out_scale (the modulation gradient of
                                ModulatedConv2d, :311-320).
    After the launch the results are in .colsum / .dot (fp32)."""

    def __init__(self, mask_src=None, mask_slope=1.0, mask_gain=1.0, want_colsum=False, dot_src=None):
        self.mask_src, self.mask_slope, self.mask_gain = mask_src, float(mask_slope), float(mask_gain)
        self.want_colsum, self.dot_src = bool(want_colsum), dot_src
        self.colsum = self.dot = self._ws = None


def dot_fusable(H, W, dtype=torch.float32):
    """The modulation-gradient dot product can ride in a convolution epilogue: every tile (<= 256 rows; Winograd: 256 2x2
    tiles) stays inside one sample.  fp32 and f16 activations alike (the sums are fp32 and taken before the store rounds)."""
    return FUSE_GRAD and dtype in (torch.float32, torch.float16) and (H * W) % 1024 == 0


def _epilogue(in_scale=None, out_scale=None, bias=None, residual=None, act=False, slope=0.2, gain=2 ** 0.5, fuse=None,
              out_bchw=None, dtype=torch.float32, out_f32=False):
    e = _lib.ConvEpilogue(_p(in_scale), _p(out_scale), _p(bias), _p(residual), 1 if act else 0, slope, gain)
    e.out_f32 = 1 if out_f32 else 0
    if fuse is not None:
        B, C, H, W = out_bchw
        for t, what in ((fuse.mask_src, "mask_src"), (fuse.dot_src, "dot_src")):
            if t is not None and (tuple(t.shape) != (B, C, H, W) or t.dtype != dtype or not t.is_contiguous(memory_format=CL)):
                raise _lib.GifHipError(f"GradFuse.{what}: expected an NHWC {dtype} tensor of shape {(B, C, H, W)}, got {tuple(t.shape)} {t.dtype}")
        e.mask_src, e.mask_slope, e.mask_gain = _p(fuse.mask_src), fuse.mask_slope, fuse.mask_gain
        e.dot_src = _p(fuse.dot_src)
        if fuse.mask_src is None and fuse.dot_src is None:
            raise _lib.GifHipError("GradFuse: nothing to fuse (the plain column sum of a gradient is ops.colsum)")
        device = fuse.mask_src.device if fuse.mask_src is not None else fuse.dot_src.device
        if fuse.dot_src is not None:
            fuse.dot = torch.empty((B, C), device=device, dtype=torch.float32)
            e.dot = fuse.dot.data_ptr()
        if fuse.want_colsum:
            fuse.colsum = torch.empty((C,), device=device, dtype=torch.float32)
            e.colsum = fuse.colsum.data_ptr()
        if fuse.dot is not None or fuse.colsum is not None:
            fuse._ws = torch.empty((_lib.load().gif_conv_epilogue_ws_floats(B * H * W, C),), device=device, dtype=torch.float32)
            e.red_ws = fuse._ws.data_ptr()
    return e


# Packed / transformed weights are a pure function of (the parameter's current contents, view geometry, layout arguments):
# the same weights are packed 2-3 times per training iteration (D runs three forwards, every backward re-packs for the data
# gradient), so the results are kept until the parameter changes.  An entry is keyed on the identity of the BASE tensor (a
# weak reference proves it is still the same object), its data pointer and in-place version counter (optimisers, copy_,
# load_state_dict and FlatAdam all bump it) and the view geometry — never on an address alone.
WEIGHT_CACHE = _lib.knob("GIF_WEIGHT_CACHE", "1") != "0"
_weight_cache = {}
_WEIGHT_CACHE_MAX = 512


def clear_weight_cache():
    """Forget every packed / transformed weight (call after writing parameters through .data, which bumps no version counter)."""
    _weight_cache.clear()


def _cached_weight_op(w, tag, build):
    if not WEIGHT_CACHE:
        return build()
    base = w._base if w._base is not None else w
    key = (id(base), tag, w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    state = (base.data_ptr(), base._version)
    hit = _weight_cache.get(key)
    if hit is not None and hit[0]() is base and hit[1] == state:
        return hit[2]
    out = build()
    if len(_weight_cache) >= _WEIGHT_CACHE_MAX:
        for k in [k for k, v in _weight_cache.items() if v[0]() is None]:
            del _weight_cache[k]
        if len(_weight_cache) >= _WEIGHT_CACHE_MAX:
            _weight_cache.clear()
    _weight_cache[key] = (weakref.ref(base), state, out)
    return out


X3_TAPDENSE = _lib.knob("GIF_X3_TAPDENSE", "1") != "0"  # tap-dense K order for 3x3 layers with 8..28 contraction channels (A/B)
X3_MIN_CIN = int(_lib.knob("GIF_X3_MIN_CIN", "24"))  # gif_conv2d_x3_eligible: >= 24 (one zero-padded 32-float K chunk)


X3_MAX_INPUT_BYTES = (1 << 32) - (1 << 26)  # the bf16x3 / f16 kernels address their input through 32-bit buffer offsets (conv_igemm.hip)


def x3_conv(dtype, cin_act: int, src=None, spec=None) -> bool:
    """fp32 conv fwd/dgrad with `cin_act` contraction channels runs on the bf16x3 kernels (mode + eligibility).  Launches the
    buffer-addressed DMA does not take (>= 4 GiB of input) stay on the native fp32 kernel."""
    if src is not None and src.numel() * src.element_size() > X3_MAX_INPUT_BYTES:
        return False
    return dtype == torch.float32 and cin_act >= X3_MIN_CIN and split_mode()


def split_mode() -> bool:
    """The fp32 contractions run on the 16-bit matrix cores (bf16x3, or f16x2 with bf16x3 for the layers f16x2 does not take)."""
    return get_fp32_mfma_mode() in ("bf16x3", "f16x2")


H2_CONV = _lib.knob("GIF_H2_CONV", "1") != "0"    # f16x2 mode: direct fwd / dgrad kernels (A/B knobs per kernel family)
H2_WGRAD = _lib.knob("GIF_H2_WGRAD", "1") != "0"  # f16x2 mode: weight-gradient kernels (direct and Winograd plane GEMMs)


H2_WINO = _lib.knob("GIF_H2_WINO", "1") != "0"    # f16x2 mode: Winograd fwd / dgrad GEMM
H2_GUARD = True  # False: f16x2 launches run without their guarded bf16x3 twin (tests only: shows what the guard protects against)


# f16x2 mode: the tap-dense launches (3x3 layers with 8..28 contraction channels) stay on the bf16x3 tap-dense kernel by default — the f16x2 form
# (gif_conv2d_*_f32h2_tapdense, tests/test_gpu_f16x2.py) measured SLOWER in the step: 200.0 vs 198.5 ms in one call (24 -> 128 at 256^2: 7.9 vs
# 6.7 ms per step; these launches are bound by their per-lane tap gathers, not by the matrix pipe, and the row tracking adds VALU work)
H2_DENSE = _lib.knob("GIF_H2_DENSE", "0") != "0"


def h2_conv(x3: bool, dense: bool) -> bool:
    """This bf16x3-eligible direct launch runs the f16x2 kernels (three f16 products under per-row scales, guarded bf16x3 fallback)."""
    return (bool(x3) or bool(dense)) and (H2_DENSE or not dense) and H2_CONV and get_fp32_mfma_mode() == "f16x2"


def x3_tapdense(dtype, cin_act: int, spec, transposed: bool, epi, cout_act: int = 64) -> bool:
    """3x3 fp32 conv with 8 <= cin_act < 32 contraction channels runs the bf16x3 kernel in its tap-dense K order (include/gif_hip.h:
    the condition-noise convs and the 24 -> C layers; no per-tap padding of K).  Not for strided data gradients (tap subsets per
    output phase) and not for modulated launches.  Measured per shape at 256^2, batch 32 (profiles/r3_conv_shapes.md vs the run
    before the mode existed): 24 -> 128 3.62 vs 4.08 ms per step, 12 -> 24 0.88 vs 1.15 (native kernel); but 8 -> 12 0.76 vs 0.66
    (native) and 24 -> 12 0.79 vs 0.67 (padded bf16x3): with <= 32 output channels the launch is bound by its gathers, which the
    dense order scatters over two pixels per 128-byte row — those two keep their old kernels."""
    if not (X3_TAPDENSE and dtype == torch.float32 and split_mode() and 8 <= cin_act < 32 and cin_act % 4 == 0
            and (spec.KH, spec.KW) == (3, 3) and not (transposed and spec.stride != 1) and epi.get("in_scale") is None):
        return False
    return cin_act >= 12 and (cout_act > 32 or cin_act < X3_MIN_CIN)


def pack_weight(w: torch.Tensor, rows_are_out: bool, cout_act: int, cin_act: int, scale: float = 1.0, dtype=torch.float32, x3=False,
                tapdense=False, h2=False):
    """Pack a canonical forward-conv weight view w[O,I,KH,KW] (any strides) into [T][RP][CP] of `dtype` (fp32 or f16), or
    (x3=True) into the pre-split bf16x3 operand [T][3][RP][CP] (bf16) of the gif_conv2d_*_f32x3 entry points.

    rows_are_out=True : rows = O, cols = I (operand of gif_conv2d_fwd)
    rows_are_out=False: rows = I, cols = O (operand of gif_conv2d_bwd_data)
    cout_act / cin_act are the channel counts of the op's output / input ACTIVATIONS (>= canonical counts).
    """
    lib = _lib.load()
    O, I, KH, KW = w.shape
    so, si, sky, skx = w.stride()
    R, C, sr, sc = (O, I, so, si) if rows_are_out else (I, O, si, so)
    assert R <= cout_act and C <= cin_act, (R, cout_act, C, cin_act)
    f16 = dtype == torch.float16

    def build():
        RP, CP = ctypes.c_int(), ctypes.c_int()
        dims = lib.gif_conv2d_pack_dims_x3 if (x3 or tapdense or h2) else (lib.gif_conv2d_pack_dims_f16 if f16 else lib.gif_conv2d_pack_dims)
        _lib.check(dims(cout_act, cin_act, ctypes.byref(RP), ctypes.byref(CP)), "pack_dims")
        if tapdense:
            steps = lib.gif_conv2d_x3_tapdense_steps(cin_act, KH, KW)
            wp = torch.empty((steps, 3, RP.value, 32), device=w.device, dtype=torch.bfloat16)
            _lib.check(lib.gif_pack_weight_f32x3_tapdense(w.data_ptr(), wp.data_ptr(), R, C, cin_act, KH, KW, RP.value, sr, sc, sky, skx,
                                                          float(scale), _stream()), "pack_weight_tapdense")
            return wp
        if h2:  # f16x2 packing: [row exponents + flag][tap][2][RP][CP] f16, an opaque byte buffer
            wp = torch.empty((lib.gif_pack_weight_f32h2_bytes(KH, KW, RP.value, CP.value),), device=w.device, dtype=torch.uint8)
            fn = lib.gif_pack_weight_f32h2
        elif x3:
            wp = torch.empty((KH * KW, 3, RP.value, CP.value), device=w.device, dtype=torch.bfloat16)
            fn = lib.gif_pack_weight_f32x3
        else:
            wp = torch.empty((KH * KW, RP.value, CP.value), device=w.device, dtype=dtype)
            fn = _fn("pack_weight", dtype)
        _lib.check(fn(w.data_ptr(), wp.data_ptr(), R, C, KH, KW, RP.value, CP.value, sr, sc, sky, skx, float(scale), _stream()),
                   "pack_weight")
        return wp

    return _cached_weight_op(w, ("pack", rows_are_out, cout_act, cin_act, float(scale), dtype, bool(x3), bool(tapdense), bool(h2)), build)


def pack_weight_h2x3(w: torch.Tensor, rows_are_out: bool, cout_act: int, cin_act: int, scale: float = 1.0, tapdense=False):
    """(wp2, wp3): the f16x2 packing and the bf16x3 packing (the guarded fallback's operand) of the same weight view, ONE launch."""
    lib = _lib.load()
    O, I, KH, KW = w.shape
    so, si, sky, skx = w.stride()
    R, C, sr, sc = (O, I, so, si) if rows_are_out else (I, O, si, so)
    assert R <= cout_act and C <= cin_act, (R, cout_act, C, cin_act)

    def build():
        RP, CP = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.gif_conv2d_pack_dims_x3(cout_act, cin_act, ctypes.byref(RP), ctypes.byref(CP)), "pack_dims")
        if tapdense:
            steps = lib.gif_conv2d_x3_tapdense_steps(cin_act, KH, KW)
            wp2 = torch.empty((lib.gif_pack_weight_f32h2_tapdense_bytes(cin_act, KH, KW, RP.value),), device=w.device, dtype=torch.uint8)
            wp3 = torch.empty((steps, 3, RP.value, 32), device=w.device, dtype=torch.bfloat16)
            _lib.check(lib.gif_pack_weight_f32h2x3_tapdense(w.data_ptr(), wp2.data_ptr(), wp3.data_ptr(), R, C, cin_act, KH, KW, RP.value, sr, sc,
                                                            sky, skx, float(scale), _stream()), "pack_weight_f32h2x3_tapdense")
            return wp2, wp3
        wp2 = torch.empty((lib.gif_pack_weight_f32h2_bytes(KH, KW, RP.value, CP.value),), device=w.device, dtype=torch.uint8)
        wp3 = torch.empty((KH * KW, 3, RP.value, CP.value), device=w.device, dtype=torch.bfloat16)
        _lib.check(lib.gif_pack_weight_f32h2x3(w.data_ptr(), wp2.data_ptr(), wp3.data_ptr(), R, C, KH, KW, RP.value, CP.value, sr, sc, sky, skx,
                                               float(scale), _stream()), "pack_weight_f32h2x3")
        return wp2, wp3

    return _cached_weight_op(w, ("pack_h2x3", rows_are_out, cout_act, cin_act, float(scale), bool(tapdense)), build)


# Winograd F(2x2,3x3) dispatch for stride-1 / pad-1 3x3 convs (conv_winograd.hip).  GIF_WINOGRAD=0 forces the direct
# implicit-GEMM kernels; below GIF_WINOGRAD_MIN_TILES 2x2 tiles the launch cannot fill the chip and the direct path wins.
WINOGRAD = _lib.knob("GIF_WINOGRAD", "1") != "0"
WINOGRAD_MIN_TILES = int(_lib.knob("GIF_WINOGRAD_MIN_TILES", "8192"))
WINOGRAD_WGRAD = _lib.knob("GIF_WINOGRAD_WGRAD", "1") != "0"
WINOGRAD_X3 = _lib.knob("GIF_WINO_X3", "1") != "0"  # bf16x3 mode: Winograd fwd/dgrad GEMMs on the bf16x3 kernel too
WINOGRAD_WGRAD_MIN_TILES = int(_lib.knob("GIF_WINOGRAD_WGRAD_MIN_TILES", "2048"))  # split-K fills the chip earlier
# Per-channel-count rule (round 6, profiles/r6_dispatch_ab.md): the smaller of the two channel counts must reach these for the
# Winograd route — forward / data gradient and weight gradient separately (the weight gradient re-uses the forward's V when both
# take the route, otherwise it runs its own input transform).  Under f16x2 the direct kernels beat the Winograd route on the
# 128-channel layers (one A/B call, alternating arms, ms per step: 184.8 / 184.4 with 0 / 0; 187.0 / 186.0 with 256 / 0; 186.4 / 185.4
# with 0 / 256; 182.5 / 182.8 with 256 / 256; 188.3 with 512 / 512; 204.3 without Winograd) — the two halves only pay together: a
# direct forward leaves the weight gradient without a V to re-use.
# (The rule belongs to the f16x2 mode: with six bf16 products or the fp32-input MFMA per product the direct kernels are 1.4-2.3 x slower than
# the Winograd route at 128 channels as well — those modes keep 0 / 0.  None = by mode; an int (tests, GIF_WINOGRAD_MIN_C /
# GIF_WINOGRAD_WGRAD_MIN_C under GIF_EXPERIMENTAL=1) overrides.)
_k = _lib.knob("GIF_WINOGRAD_MIN_C", "")
WINOGRAD_MIN_C = int(_k) if _k != "" else None
_k = _lib.knob("GIF_WINOGRAD_WGRAD_MIN_C", "")
WINOGRAD_WGRAD_MIN_C = int(_k) if _k != "" else None
del _k
WINOGRAD_MIN_C_F16X2 = 256


def _winograd_min_c(override):
    if override is not None:
        return override
    return WINOGRAD_MIN_C_F16X2 if get_fp32_mfma_mode() == "f16x2" else 0


_winograd_calls = 0


def prof_winograd_calls():
    """Number of Winograd launches so far (tests use it to prove the path under test actually ran)."""
    return _winograd_calls


def winograd_eligible(spec: ConvSpec, B, H, W, cin_act, cout_act=64, min_tiles=None, dtype=torch.float32, min_c=None):
    # cout < 48 wastes over a quarter of the GEMM's 64-wide N tile; the direct 256x32 kernel is faster there (measured)
    min_tiles = WINOGRAD_MIN_TILES if min_tiles is None else min_tiles
    min_c = _winograd_min_c(WINOGRAD_MIN_C) if min_c is None else min_c
    return (WINOGRAD and dtype == torch.float32 and tuple(spec) == (3, 3, 1, 1) and H % 2 == 0 and W % 2 == 0 and cin_act >= 32 and cout_act >= 48
            and min(cin_act, cout_act) >= min_c and B * (H // 2) * (W // 2) >= min_tiles)


def conv3x3_winograd(x, w, rows_are_out: bool, cout_act: int, wscale=1.0, keep_v=False, **epi):
    """act(out_scale * conv3x3_s1_p1(in_scale * x, w) + residual + bias) via Winograd F(2x2,3x3).

    rows_are_out=True : forward conv with w[O,I,3,3];  False: its data gradient (taps rotated, channels swapped).
    keep_v=True returns (y, V): V = the transformed (in_scale * x), reusable by conv3x3_winograd_wgrad(big=x)."""
    global _winograd_calls
    _winograd_calls += 1
    lib = _lib.load()
    x = nhwc(x)
    B, C, H, W = x.shape
    O, I = w.shape[:2]
    so, si, sky, skx = w.stride()
    R, Cc, sr, sc = (O, I, so, si) if rows_are_out else (I, O, si, so)
    assert R <= cout_act and Cc <= C, (R, cout_act, Cc, C)

    # bf16x3: the GEMM's 128-wide N tile wants full tiles; other channel counts stay on the native GEMM (GIF_WINO_X3=0: A/B)
    x3 = WINOGRAD_X3 and split_mode() and cout_act % 128 == 0

    h2 = x3 and H2_WINO and get_fp32_mfma_mode() == "f16x2"

    def build():
        RP, CP = ctypes.c_int(), ctypes.c_int()
        dims = lib.gif_winograd_pack_dims_x3 if x3 else lib.gif_winograd_pack_dims
        _lib.check(dims(cout_act, C, ctypes.byref(RP), ctypes.byref(CP)), "winograd_pack_dims")
        if h2:  # the f16x2 transform and the bf16x3 transform (the guarded fallback's operand) in one launch
            U2 = torch.empty((lib.gif_winograd_weight_f32h2_bytes(RP.value, CP.value),), device=x.device, dtype=torch.uint8)
            U3 = torch.empty((16, 3, RP.value, CP.value), device=x.device, dtype=torch.bfloat16)
            _lib.check(lib.gif_winograd_weight_f32h2(w.data_ptr(), U2.data_ptr(), U3.data_ptr(), R, Cc, RP.value, CP.value, sr, sc, sky, skx,
                                                     0 if rows_are_out else 1, float(wscale), _stream()), "winograd_weight_f32h2")
            return U2, U3
        if x3:
            U = torch.empty((16, 3, RP.value, CP.value), device=x.device, dtype=torch.bfloat16)
            fn = lib.gif_winograd_weight_f32x3
        else:
            U = torch.empty((16, RP.value, CP.value), device=x.device, dtype=torch.float32)
            fn = lib.gif_winograd_weight_f32
        _lib.check(fn(w.data_ptr(), U.data_ptr(), R, Cc, RP.value, CP.value, sr, sc, sky, skx, 0 if rows_are_out else 1, float(wscale),
                      _stream()), "winograd_weight")
        return U

    U = _cached_weight_op(w, ("wino", rows_are_out, cout_act, C, float(wscale), x3, h2), build)
    V = torch.empty((lib.gif_winograd_workspace_floats(B, H, W, C),), device=x.device, dtype=torch.float32)
    out = empty_nhwc(B, cout_act, H, W, x.device)
    e = _epilogue(out_bchw=(B, cout_act, H, W), **epi)
    if h2:  # f16x2 GEMM + its guarded bf16x3 twin
        U2, U3 = U
        _lib.check(lib.gif_conv3x3_winograd_f32h2(x.data_ptr(), U2.data_ptr(), U3.data_ptr() if H2_GUARD else None, out.data_ptr(),
                                                  V.data_ptr(), B, H, W, C, cout_act, ctypes.byref(e), _stream()), "conv3x3_winograd_f32h2")
        return (out, V) if keep_v else out
    fn = lib.gif_conv3x3_winograd_f32x3 if x3 else lib.gif_conv3x3_winograd_f32
    _lib.check(fn(x.data_ptr(), U.data_ptr(), out.data_ptr(), V.data_ptr(), B, H, W, C, cout_act, ctypes.byref(e), _stream()),
               "conv3x3_winograd")
    return (out, V) if keep_v else out


def _epi_check(x, epi):
    r = epi.get("residual")
    if r is not None:
        _same_dtype(x, r, "conv epilogue residual")


def conv_fwd(big, w, spec: ConvSpec, wscale=1.0, keep_v=False, **epi):
    """small = conv2d(big, w[O,I,KH,KW]) ; returns [B, cpad(O), Hs, Ws] in big's dtype.

    keep_v=True returns (small, V) where V is the Winograd-transformed (in_scale * big) if that path ran, else None;
    pass it to conv_wgrad(big_v=V) for the weight gradient of the same (big, in_scale)."""
    big = nhwc(big)
    dt = big.dtype
    B, Cb, Hb, Wb = big.shape
    O = w.shape[0]
    Cs = cpad(O, dt)
    Hs, Ws = spec.small_hw(Hb, Wb)
    _epi_check(big, epi)
    if dt != torch.float16:
        epi.pop("out_f32", None)  # (fp32 activations: the result is fp32 anyway)
    if winograd_eligible(spec, B, Hb, Wb, Cb, Cs, dtype=dt):
        return conv3x3_winograd(big, w, True, Cs, wscale, keep_v=keep_v, **epi)
    if keep_v:
        return conv_fwd(big, w, spec, wscale, **epi), None
    x3 = x3_conv(dt, Cb, big, spec)
    dense = x3_tapdense(dt, Cb, spec, False, epi, Cs) and big.numel() * 4 <= X3_MAX_INPUT_BYTES
    h2 = h2_conv(x3, dense)
    wp = None if h2 else pack_weight(w, True, Cs, Cb, wscale, dt, x3=x3, tapdense=dense)
    # out_f32 (f16 activations only): fp32 result, e.g. the RGB image of ToRGB
    out = empty_nhwc(B, Cs, Hs, Ws, big.device, torch.float32 if epi.get("out_f32") else dt)
    g = _geom(B, Hb, Wb, Cb, Hs, Ws, Cs, spec)
    e = _epilogue(out_bchw=(B, Cs, Hs, Ws), dtype=dt, **epi)
    if h2:  # f16x2 kernels + the bf16x3 packing for their guarded fallback
        wp2, wp = pack_weight_h2x3(w, True, Cs, Cb, wscale, tapdense=dense)
        _lib.check((_lib.load().gif_conv2d_fwd_f32h2_tapdense if dense else _lib.load().gif_conv2d_fwd_f32h2)(big.data_ptr(), wp2.data_ptr(), wp.data_ptr() if H2_GUARD else None, out.data_ptr(), ctypes.byref(g),
                                                    ctypes.byref(e), _stream()), "conv2d_fwd_f32h2")
        return out
    fn = _lib.load().gif_conv2d_fwd_f32x3_tapdense if dense else _lib.load().gif_conv2d_fwd_f32x3 if x3 else _fn("conv2d_fwd", dt)
    _lib.check(fn(big.data_ptr(), wp.data_ptr(), out.data_ptr(), ctypes.byref(g), ctypes.byref(e), _stream()), "conv2d_fwd")
    return out


def conv_bwd_data(small, w, spec: ConvSpec, big_hw, wscale=1.0, **epi):
    """big = conv_transpose2d(small, w[O,I,KH,KW]) ; returns [B, cpad(I), Hb, Wb] in small's dtype."""
    small = nhwc(small)
    dt = small.dtype
    B, Cs, Hs, Ws = small.shape
    I = w.shape[1]
    Cb = cpad(I, dt)
    Hb, Wb = big_hw
    _epi_check(small, epi)
    if (Hb, Wb) == (Hs, Ws) and winograd_eligible(spec, B, Hs, Ws, Cs, Cb, dtype=dt):
        return conv3x3_winograd(small, w, False, Cb, wscale, **epi)
    x3 = x3_conv(dt, Cs, small, spec)
    dense = x3_tapdense(dt, Cs, spec, True, epi, Cb) and small.numel() * 4 <= X3_MAX_INPUT_BYTES
    h2 = h2_conv(x3, dense)
    wp = None if h2 else pack_weight(w, False, Cb, Cs, wscale, dt, x3=x3, tapdense=dense)
    out = empty_nhwc(B, Cb, Hb, Wb, small.device, dt)
    g = _geom(B, Hb, Wb, Cb, Hs, Ws, Cs, spec)
    e = _epilogue(out_bchw=(B, Cb, Hb, Wb), dtype=dt, **epi)
    if h2:
        wp2, wp = pack_weight_h2x3(w, False, Cb, Cs, wscale, tapdense=dense)
        _lib.check((_lib.load().gif_conv2d_bwd_data_f32h2_tapdense if dense else _lib.load().gif_conv2d_bwd_data_f32h2)(small.data_ptr(), wp2.data_ptr(), wp.data_ptr() if H2_GUARD else None, out.data_ptr(), ctypes.byref(g),
                                                         ctypes.byref(e), _stream()), "conv2d_bwd_data_f32h2")
        return out
    fn = (_lib.load().gif_conv2d_bwd_data_f32x3_tapdense if dense else _lib.load().gif_conv2d_bwd_data_f32x3 if x3
          else _fn("conv2d_bwd_data", dt))
    _lib.check(fn(small.data_ptr(), wp.data_ptr(), out.data_ptr(), ctypes.byref(g), ctypes.byref(e), _stream()), "conv2d_bwd_data")
    return out


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def conv3x3_winograd_wgrad(small, big, O, I, wscale=1.0, small_scale=None, big_scale=None, big_v=None):
    """dW[O,I,3,3] of the stride-1 / pad-1 3x3 conv via Winograd F(3x3,2x2) (transforms + 16 plane GEMMs + unpack).
    big_v: the V kept from the forward pass over the same (big, big_scale) — skips the input transform."""
    global _winograd_calls
    _winograd_calls += 1
    lib = _lib.load()
    small, big = nhwc(small), nhwc(big)
    B, Cs, H, W = small.shape
    Cb = big.shape[1]
    assert big.shape == (B, Cb, H, W) and O <= Cs and I <= Cb
    RP, CP = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.gif_conv2d_wgrad_dims(pad32(Cs), pad32(Cb), ctypes.byref(RP), ctypes.byref(CP)), "wgrad_dims")
    nsplit = lib.gif_conv3x3_winograd_wgrad_splits(B, H, W, Cs, Cb)
    dev = small.device
    nv = lib.gif_winograd_workspace_floats(B, H, W, Cb)
    if big_v is not None and big_v.numel() != nv:
        raise _lib.GifHipError(f"conv3x3_winograd_wgrad: cached V has {big_v.numel()} floats, expected {nv}")
    V = big_v if big_v is not None else torch.empty((nv,), device=dev, dtype=torch.float32)
    Mg = torch.empty((lib.gif_winograd_workspace_floats(B, H, W, Cs),), device=dev, dtype=torch.float32)
    ws = torch.empty((nsplit, 16, RP.value, CP.value), device=dev, dtype=torch.float32)
    fn = (lib.gif_conv3x3_winograd_wgrad_f32h2 if (H2_WGRAD and get_fp32_mfma_mode() == "f16x2") else
          lib.gif_conv3x3_winograd_wgrad_f32x3 if split_mode() else lib.gif_conv3x3_winograd_wgrad_f32)
    _lib.check(fn(None if big_v is not None else big.data_ptr(), small.data_ptr(), V.data_ptr(), Mg.data_ptr(), ws.data_ptr(),
                                                  _p(small_scale), _p(big_scale), B, H, W, Cs, Cb, nsplit, _stream()),
               "conv3x3_winograd_wgrad")
    dw = torch.empty((O, I, 3, 3), device=dev, dtype=torch.float32)
    so, si, sky, skx = dw.stride()
    _lib.check(lib.gif_winograd_unpack_wgrad_f32(ws.data_ptr(), dw.data_ptr(), nsplit, O, I, RP.value, CP.value, so, si, sky,
                                                 skx, float(wscale), _stream()), "winograd_unpack_wgrad")
    return dw


def conv_wgrad(small, big, spec: ConvSpec, O, I, wscale=1.0, small_scale=None, big_scale=None, big_v=None):
    """dW[O,I,KH,KW] = wscale * sum small (x) big  (contiguous canonical layout, fp32 for fp32 AND f16 operands).
    big_v: optional Winograd V of (big_scale * big) kept from conv_fwd(keep_v=True)."""
    lib = _lib.load()
    small, big = nhwc(small), nhwc(big)
    _same_dtype(small, big, "conv_wgrad")
    dt = small.dtype
    f16 = dt == torch.float16
    B, Cs, Hs, Ws = small.shape
    _, Cb, Hb, Wb = big.shape
    assert O <= Cs and I <= Cb
    if (WINOGRAD_WGRAD and (Hb, Wb) == (Hs, Ws) and Cs >= 64 and Cb >= 64
            and winograd_eligible(spec, B, Hs, Ws, Cb, Cs, min(WINOGRAD_MIN_TILES, WINOGRAD_WGRAD_MIN_TILES), dtype=dt,
                                  min_c=_winograd_min_c(WINOGRAD_WGRAD_MIN_C))):
        return conv3x3_winograd_wgrad(small, big, O, I, wscale, small_scale, big_scale, big_v)
    g = _geom(B, Hb, Wb, Cb, Hs, Ws, Cs, spec)
    RP, CP = ctypes.c_int(), ctypes.c_int()
    dims, splits = ((lib.gif_conv2d_wgrad_dims_f16, lib.gif_conv2d_wgrad_splits_f16) if f16 else
                    (lib.gif_conv2d_wgrad_dims, lib.gif_conv2d_wgrad_splits))
    _lib.check(dims(Cs, Cb, ctypes.byref(RP), ctypes.byref(CP)), "wgrad_dims")
    nsplit = splits(ctypes.byref(g))
    T = spec.KH * spec.KW
    ws = torch.empty((nsplit, T, RP.value, CP.value), device=small.device, dtype=torch.float32)
    fn = (lib.gif_conv2d_wgrad_f32h2 if (not f16 and H2_WGRAD and get_fp32_mfma_mode() == "f16x2") else
          lib.gif_conv2d_wgrad_f32x3 if (not f16 and split_mode()) else _fn("conv2d_wgrad", dt))
    _lib.check(fn(small.data_ptr(), big.data_ptr(), ws.data_ptr(), _p(small_scale), _p(big_scale), ctypes.byref(g), nsplit, _stream()),
               "conv2d_wgrad")
    dw = torch.empty((O, I, spec.KH, spec.KW), device=small.device, dtype=torch.float32)
    so, si, sky, skx = dw.stride()
    _lib.check(lib.gif_unpack_wgrad_f32(ws.data_ptr(), dw.data_ptr(), nsplit, O, I, spec.KH, spec.KW, RP.value, CP.value,
                                        so, si, sky, skx, float(wscale), _stream()), "unpack_wgrad")
    return dw


def fir_fusable(C, up, down, kshape, dtype=torch.float32):
    """The blur kernels (4x4 FIR, up = down = 1) take the mask / column-sum fusions for power-of-two channel counts."""
    return (FUSE_GRAD and dtype in (torch.float32, torch.float16) and up == 1 and down == 1 and tuple(kshape) == (4, 4)
            and C & (C - 1) == 0 and C <= 1024)


def upfirdn2d(x, k, up, down, pad0, out_hw, flip=True, bias=None, residual=None, act=False, slope=0.2, gain=2 ** 0.5, fuse=None):
    x = nhwc(x)
    B, C, Hi, Wi = x.shape
    Ho, Wo = out_hw
    KH, KW = k.shape
    k = k.contiguous()
    if residual is not None:
        _same_dtype(x, residual, "upfirdn2d residual")
    y = empty_nhwc(B, C, Ho, Wo, x.device, x.dtype)
    e = _epilogue(bias=bias, residual=residual, act=act, slope=slope, gain=gain, fuse=fuse, out_bchw=(B, C, Ho, Wo), dtype=x.dtype)
    _lib.check(_fn("upfirdn2d", x.dtype)(x.data_ptr(), k.data_ptr(), y.data_ptr(), B, Hi, Wi, C, Ho, Wo, up, down, pad0, pad0,
                                         KH, KW, 1 if flip else 0, ctypes.byref(e), _stream()), "upfirdn2d")
    return y


def bias_act(x, bias=None, residual=None, slope=0.2, gain=2 ** 0.5):
    x = nhwc(x)
    B, C, H, W = x.shape
    if residual is not None:
        residual = nhwc(residual)
        assert residual.shape == x.shape
        _same_dtype(x, residual, "bias_act residual")
    y = empty_nhwc(B, C, H, W, x.device, x.dtype)
    _lib.check(_fn("bias_act", x.dtype)(x.data_ptr(), _p(bias), _p(residual), y.data_ptr(), B * H * W, C, slope, gain,
                                        _stream()), "bias_act")
    return y


def bias_act_bwd(gy, y, want_gbias, slope=0.2, gain=2 ** 0.5):
    lib = _lib.load()
    gy, y = nhwc(gy), nhwc(y)
    _same_dtype(gy, y, "bias_act_bwd")
    B, C, H, W = y.shape
    npix = B * H * W
    gx = empty_nhwc(B, C, H, W, y.device, y.dtype)
    gbias = partial = None
    if want_gbias:
        gbias = torch.empty((C,), device=y.device, dtype=torch.float32)
        partial = torch.empty((lib.gif_colsum_partial_floats(npix, C),), device=y.device, dtype=torch.float32)
    _lib.check(_fn("bias_act_bwd", y.dtype)(gy.data_ptr(), y.data_ptr(), gx.data_ptr(), _p(gbias), _p(partial), npix, C, slope,
                                            gain, _stream()), "bias_act_bwd")
    return gx, gbias


def colsum(x):
    """[B,C,H,W] (NHWC) -> [C] sum over B,H,W (fp32)."""
    lib = _lib.load()
    x = nhwc(x)
    B, C, H, W = x.shape
    npix = B * H * W
    out = torch.empty((C,), device=x.device, dtype=torch.float32)
    partial = torch.empty((lib.gif_colsum_partial_floats(npix, C),), device=x.device, dtype=torch.float32)
    _lib.check(_fn("colsum", x.dtype)(x.data_ptr(), out.data_ptr(), partial.data_ptr(), npix, C, _stream()), "colsum")
    return out


def mul_reduce(a, b, scale=None, want_scaled=False):
    """out[b,c] = sum_hw a*b (fp32) ; optionally scaled = scale[b,c]*a (activation dtype)."""
    lib = _lib.load()
    a, b = nhwc(a), nhwc(b)
    _same_dtype(a, b, "mul_reduce")
    B, C, H, W = a.shape
    assert b.shape == a.shape
    nchunk = lib.gif_mul_reduce_chunks(H * W)
    out = torch.empty((B, C), device=a.device, dtype=torch.float32)
    partial = torch.empty((B * nchunk * C,), device=a.device, dtype=torch.float32)
    scaled = empty_nhwc(B, C, H, W, a.device, a.dtype) if want_scaled else None
    if scale is not None:
        scale = scale.contiguous()
        assert scale.shape == (B, C) and scale.dtype == torch.float32
    _lib.check(_fn("mul_reduce", a.dtype)(a.data_ptr(), b.data_ptr(), _p(scale), _p(scaled), out.data_ptr(), partial.data_ptr(),
                                          B, H * W, C, _stream()), "mul_reduce")
    return out, scaled


def bilinear_down(x, S, backward_to=None):
    """Forward: x [B,C,R,R] -> [B,C,S,S].  backward_to=R: x is the gradient of a level, returns the gradient [B,C,R,R]."""
    lib = _lib.load()
    x = nhwc(x)
    if x.dtype != torch.float32:
        raise _lib.GifHipError("bilinear_down is fp32 only (the condition pyramid is built in fp32 and cast per level)")
    B, C = x.shape[:2]
    if backward_to is None:
        R = x.shape[2]
        y = empty_nhwc(B, C, S, S, x.device)
        _lib.check(lib.gif_bilinear_down_f32(x.data_ptr(), y.data_ptr(), B, R, S, C, 0, _stream()), "bilinear_down")
    else:
        R = backward_to
        y = empty_nhwc(B, C, R, R, x.device)
        _lib.check(lib.gif_bilinear_down_f32(x.data_ptr(), y.data_ptr(), B, R, S, C, 1, _stream()), "bilinear_down_bwd")
    return y


def _i64x4(t):
    return (ctypes.c_int64 * 4)(*t.stride())


@_device_guard
def pack_nhwc(src0, off0, src1, off1, cp, dtype):
    """[B,cp,H,W] NHWC tensor of `dtype` (fp32 / f16) with channels [off0, off0 + C0) = src0, [off1, off1 + C1) = src1 (or None),
    zeros elsewhere — torch.cat + channel padding + conversion + layout change in one pass (include/gif_hip.h).  Sources: fp32,
    logical [B,C,H,W], any strides."""
    lib = _lib.load()
    for t in (src0, src1):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4):
            raise _lib.GifHipError("pack_nhwc needs 4-D fp32 device tensors (no CPU fallback)")
    B, C0, H, W = src0.shape
    if src1 is not None and (src1.shape[0], src1.shape[2], src1.shape[3]) != (B, H, W):
        raise _lib.GifHipError(f"pack_nhwc: sources disagree: {tuple(src0.shape)} vs {tuple(src1.shape)}")
    out = empty_nhwc(B, cp, H, W, src0.device, dtype)
    _lib.check(_fn("pack_nhwc", dtype)(src0.data_ptr(), C0, off0, _i64x4(src0), _p(src1), 0 if src1 is None else src1.shape[1], off1,
                                       None if src1 is None else _i64x4(src1), out.data_ptr(), B, H, W, cp, _stream()), "pack_nhwc")
    return out


@_device_guard
def unpack_nhwc(g, c_off, C):
    """Adjoint of pack_nhwc w.r.t. one source: channels [c_off, c_off + C) of the NHWC tensor g -> fp32 [B,C,H,W] (channels_last)."""
    g = nhwc(g)
    B, cp, H, W = g.shape
    out = torch.empty((B, C, H, W), device=g.device, dtype=torch.float32, memory_format=CL)
    _lib.check(_fn("unpack_nhwc", g.dtype)(g.data_ptr(), out.data_ptr(), B, H, W, cp, c_off, C, _stream()), "unpack_nhwc")
    return out


def act_inv_mul_reduce(g, y, residual, bias, slope, gain):
    """out[b,c] = sum_hw g * (act^-1(y) - residual - bias[c])  (see gif_hip.h)."""
    lib = _lib.load()
    g, y = nhwc(g), nhwc(y)
    _same_dtype(g, y, "act_inv_mul_reduce")
    B, C, H, W = y.shape
    nchunk = lib.gif_mul_reduce_chunks(H * W)
    out = torch.empty((B, C), device=y.device, dtype=torch.float32)
    partial = torch.empty((B * nchunk * C,), device=y.device, dtype=torch.float32)
    _lib.check(_fn("act_inv_mul_reduce", y.dtype)(g.data_ptr(), y.data_ptr(), _p(residual), _p(bias), out.data_ptr(),
                                                  partial.data_ptr(), B, H * W, C, slope, gain, _stream()),
               "act_inv_mul_reduce")
    return out


def mbstd_fwd(x, G, Cy):
    lib = _lib.load()
    x = nhwc(x)
    if x.dtype != torch.float32:
        raise _lib.GifHipError("minibatch stddev is fp32 only (functional.minibatch_stddev casts the 4x4 tensor)")
    B, C, H, W = x.shape
    y = empty_nhwc(B, Cy, H, W, x.device)
    stat = torch.empty((B // G,), device=x.device, dtype=torch.float32)
    _lib.check(lib.gif_mbstd_fwd_f32(x.data_ptr(), y.data_ptr(), stat.data_ptr(), B, H, W, C, Cy, G, _stream()), "mbstd_fwd")
    return y, stat


def mbstd_bwd(x, gy, G):
    lib = _lib.load()
    x, gy = nhwc(x), nhwc(gy)
    B, C, H, W = x.shape
    Cy = gy.shape[1]
    gx = empty_nhwc(B, C, H, W, x.device)
    _lib.check(lib.gif_mbstd_bwd_f32(x.data_ptr(), gy.data_ptr(), gx.data_ptr(), B, H, W, C, Cy, G, _stream()), "mbstd_bwd")
    return gx


def sqnorm_per_sample(g):
    lib = _lib.load()
    if not g.is_cuda or g.dtype != torch.float32:
        raise _lib.GifHipError("sqnorm_per_sample needs an fp32 device tensor")
    g = g if g.is_contiguous() or g.is_contiguous(memory_format=CL) else g.contiguous()
    B = g.shape[0]
    out = torch.empty((B,), device=g.device, dtype=torch.float32)
    _lib.check(lib.gif_sqnorm_per_sample_f32(g.data_ptr(), out.data_ptr(), B, g.numel() // max(B, 1), _stream()), "sqnorm")
    return out


# One zero-initialised rasteriser workspace per (device, stream, problem size), kept across calls: the tile kernel hands the
# counters back zeroed, so the per-call memset node is switched off for exactly these pointers
# (gif_rasterize_assume_clean_workspace(ws, 1)).  A failed call drops its entry (the counters may be dirty).
_raster_ws = {}
_RASTER_WS_MAX = 8


def _raster_ws_drop(lib, key):
    ws = _raster_ws.pop(key, None)
    if ws is not None:
        lib.gif_rasterize_assume_clean_workspace(ws.data_ptr(), 0)  # before the memory can be handed to somebody else


def _raster_workspace(lib, dev, B, F, h, w):
    key = (dev.index, torch.cuda.current_stream().cuda_stream, B, F, h, w)
    ws = _raster_ws.get(key)
    if ws is None:
        if len(_raster_ws) >= _RASTER_WS_MAX:
            _raster_ws_drop(lib, next(iter(_raster_ws)))
        ws = torch.zeros((max(lib.gif_rasterize_workspace_bytes(B, F, h, w) // 8, 1),), device=dev, dtype=torch.int64)
        _raster_ws[key] = ws
        _lib.check(lib.gif_rasterize_assume_clean_workspace(ws.data_ptr(), 1), "rasterize_assume_clean_workspace")
    return key, ws


def rasterize(face_vertices, depth, tri, out3, h, w, face_colors=None):
    """float32 or float64 buffers (all floating tensors of one dtype, like the reference's AT_DISPATCH_FLOATING_TYPES)."""
    lib = _lib.load()
    B, F = face_vertices.shape[:2]
    f64 = face_vertices.dtype == torch.float64
    key, ws = _raster_workspace(lib, face_vertices.device, B, F, h, w)
    plain, colors = (lib.gif_rasterize_f64, lib.gif_rasterize_colors_f64) if f64 else (lib.gif_rasterize_f32,
                                                                                        lib.gif_rasterize_colors_f32)
    if face_colors is None:
        rc = plain(face_vertices.data_ptr(), depth.data_ptr(), tri.data_ptr(), out3.data_ptr(), B, F, h, w, ws.data_ptr(),
                   _stream())
    else:
        rc = colors(face_vertices.data_ptr(), face_colors.data_ptr(), depth.data_ptr(), tri.data_ptr(), out3.data_ptr(), B, F,
                    h, w, ws.data_ptr(), _stream())
    if rc != 0:
        _raster_ws_drop(lib, key)  # (the library already forgot the pointer: its counters may be dirty)
    _lib.check(rc, "rasterize")


def _mask_u8(m, hw):
    if m is None:
        return None
    m = m.reshape(-1)
    if m.numel() != hw:
        raise _lib.GifHipError(f"texture_pair_loss: visibility mask has {m.numel()} elements, expected {hw}")
    return (m != 0).to(torch.uint8).contiguous()


def texture_pair_loss(a, b, ma, mb, f, gloss=None):
    """mean(sigmoid(((a-b)*ma*mb)^2) * f) over [C,H,W]; with gloss (device scalar) returns d/da instead (d/db = -d/da)."""
    lib = _lib.load()
    if not (a.is_cuda and b.is_cuda and f.is_cuda) or a.dtype != torch.float32 or a.dim() != 3 or a.shape != b.shape:
        raise _lib.GifHipError("texture_pair_loss: need two fp32 device textures [C,H,W] of the same shape (no CPU fallback)")
    a, b = a.contiguous(), b.contiguous()
    C, H, W = a.shape
    HW = H * W
    f = f.to(torch.float32).reshape(-1).contiguous()
    if f.numel() != HW:
        raise _lib.GifHipError(f"texture_pair_loss: face mask has {f.numel()} elements, expected {HW}")
    ma, mb = _mask_u8(ma, HW), _mask_u8(mb, HW)
    if gloss is None:
        partial = torch.empty((lib.gif_texture_pair_loss_partials(C * HW),), device=a.device, dtype=torch.float32)
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        _lib.check(lib.gif_texture_pair_loss_f32(a.data_ptr(), b.data_ptr(), _p(ma), _p(mb), f.data_ptr(), partial.data_ptr(),
                                                 loss.data_ptr(), C, HW, _stream()), "texture_pair_loss")
        return loss
    ga = torch.empty_like(a)
    gloss = gloss.to(torch.float32).contiguous()
    _lib.check(lib.gif_texture_pair_loss_bwd_f32(a.data_ptr(), b.data_ptr(), _p(ma), _p(mb), f.data_ptr(), gloss.data_ptr(),
                                                 ga.data_ptr(), C, HW, _stream()), "texture_pair_loss_bwd")
    return ga


def resize(x, out_hw, mode: str, backward_to=None):
    """NCHW fp32 resize (bilinear / bicubic, align_corners=False).  backward_to=(Hi, Wi): apply the adjoint to x instead."""
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4:
        raise _lib.GifHipError("resize: need a 4-D fp32 device tensor (no CPU fallback)")
    if mode not in ("bilinear", "bicubic"):
        raise _lib.GifHipError(f"resize: mode {mode!r} not supported (bilinear | bicubic)")
    x = x.contiguous()
    B, C, H, W = x.shape
    m = 0 if mode == "bilinear" else 1
    if backward_to is None:
        Ho, Wo = out_hw
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32)
        _lib.check(lib.gif_resize_f32(x.data_ptr(), y.data_ptr(), B * C, H, W, Ho, Wo, m, _stream()), "resize")
        return y
    Hi, Wi = backward_to
    gx = torch.empty((B, C, Hi, Wi), device=x.device, dtype=torch.float32)
    _lib.check(lib.gif_resize_bwd_f32(x.data_ptr(), gx.data_ptr(), B * C, Hi, Wi, H, W, m, _stream()), "resize_bwd")
    return gx


for _name in ("pack_weight", "conv3x3_winograd", "conv_fwd", "conv_bwd_data", "conv3x3_winograd_wgrad", "conv_wgrad", "upfirdn2d",
              "bias_act", "bias_act_bwd", "colsum", "mul_reduce", "bilinear_down", "act_inv_mul_reduce", "mbstd_fwd", "mbstd_bwd",
              "sqnorm_per_sample", "rasterize", "texture_pair_loss", "resize"):
    globals()[_name] = _device_guard(globals()[_name])
del _name


def _mat(t, what):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
        raise _lib.GifHipError(f"{what}: need a 2-D fp32 device tensor (no CPU fallback), got {tuple(t.shape)} {t.dtype} on {t.device}")
    return t if t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 else t.contiguous()


def linear_nt(a, b, bias=None, scale=1.0, act=False, slope=0.2, gain=1.0, n_pad=None):
    """[M, n_pad] = act(scale * a[M,K] @ b[N,K]^T + bias[n_pad]); columns N..n_pad are zero.  K % 4 == 0."""
    lib = _lib.load()
    a, b = _mat(a, "linear_nt"), _mat(b, "linear_nt")
    M = a.shape[0]
    N, K = b.shape
    assert a.shape[1] >= K, (a.shape, b.shape)  # `a` may carry padding columns beyond K
    n_pad = N if n_pad is None else n_pad
    c = torch.empty((M, n_pad), device=a.device, dtype=torch.float32)
    _lib.check(lib.gif_linear_nt_f32(a.data_ptr(), b.data_ptr(), _p(bias), c.data_ptr(), M, N, K, a.stride(0), b.stride(0), n_pad,
                                     n_pad, float(scale), 1 if act else 0, float(slope), float(gain), _stream()), "linear_nt")
    return c


LINEAR_BANK_MAX = 40  # segments per launch (kernel-argument table, csrc/linear.hip)


def linear_bank_ok(x, weights):
    """The modulation bank takes fp32 [n, K] weights with n % 8 == 0 over one [M, K'] input (K % 4 == 0)."""
    K = weights[0].shape[1]
    return (0 < len(weights) <= LINEAR_BANK_MAX and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] >= K and K % 4 == 0
            and all(w.dim() == 2 and w.shape[1] == K and w.shape[0] % 8 == 0 and w.dtype == torch.float32 for w in weights))


def _bank_segs(weights):
    segs = (_lib.LinearBankSeg * len(weights))()
    keep = []
    for g, w in zip(segs, weights):
        w = _mat(w, "linear_bank")
        assert w.stride(0) == w.shape[1], "linear_bank: weight rows must be contiguous"
        keep.append(w)
        g.w, g.n = w.data_ptr(), w.shape[0]
    return segs, keep


def linear_bank_fwd(x, weights, biases, scale):
    """[scale * x @ w_l^T + b_l for l]: ONE launch for all layers (gif_linear_bank_fwd_f32)."""
    lib = _lib.load()
    x = _mat(x, "linear_bank_fwd")
    M, K = x.shape[0], weights[0].shape[1]
    segs, keep = _bank_segs(weights)
    outs = []
    for g, w, b in zip(segs, weights, biases):
        o = torch.empty((M, w.shape[0]), device=x.device, dtype=torch.float32)
        if b is not None:
            assert b.dtype == torch.float32 and b.numel() == w.shape[0] and b.is_contiguous()
            g.bias = b.data_ptr()
        g.s = o.data_ptr()
        outs.append(o)
    _lib.check(lib.gif_linear_bank_fwd_f32(x.data_ptr(), M, K, x.stride(0), segs, len(weights), float(scale), _stream()), "linear_bank_fwd")
    return outs


def linear_bank_bwd(x, weights, grads, scale, want_x, want_w, want_b, x_cols=None):
    """(gx [M, x_cols] or None, [gw_l] or None, [gb_l] or None) of linear_bank_fwd for the per-layer output gradients `grads`:
    one launch for all weight / bias gradients, one for gx (gif_linear_bank_bwd_f32)."""
    lib = _lib.load()
    x = _mat(x, "linear_bank_bwd")
    M, K = x.shape[0], weights[0].shape[1]
    segs, keep = _bank_segs(weights)
    gws = gbs = None
    new = torch.zeros_like if M == 0 else torch.empty_like  # (an empty batch launches nothing: the gradients are zeros, not garbage)
    if want_w:
        gws = [new(w, memory_format=torch.contiguous_format) for w in weights]
    if want_b:
        gbs = [(torch.zeros if M == 0 else torch.empty)((w.shape[0],), device=x.device, dtype=torch.float32) for w in weights]
    if want_b and not want_w:  # the column sums ride on the weight-gradient launch
        gws = [new(w, memory_format=torch.contiguous_format) for w in weights]
    for i, (g, w, gs) in enumerate(zip(segs, weights, grads)):
        gs = gs.contiguous()
        assert gs.shape == (M, w.shape[0]) and gs.dtype == torch.float32, (gs.shape, gs.dtype)
        keep.append(gs)
        g.gs = gs.data_ptr()
        if gws is not None:
            g.gw = gws[i].data_ptr()
        if gbs is not None:
            g.gbias = gbs[i].data_ptr()
    gx = None
    x_cols = K if x_cols is None else x_cols
    if want_x:
        gx = torch.empty((M, x_cols), device=x.device, dtype=torch.float32)
    _lib.check(lib.gif_linear_bank_bwd_f32(x.data_ptr(), M, K, x.stride(0), segs, len(weights), float(scale), _p(gx),
                                           x_cols, x_cols, _stream()), "linear_bank_bwd")
    return gx, (gws if want_w else None), gbs


def linear_nn(a, b, scale=1.0, n_valid=None, k_pad=None):
    """[M, k_pad] = scale * a[M, :n_valid] @ b[n_valid, K]; columns K..k_pad are zero."""
    lib = _lib.load()
    a, b = _mat(a, "linear_nn"), _mat(b, "linear_nn")
    M = a.shape[0]
    N, K = b.shape
    assert a.shape[1] >= N
    k_pad = K if k_pad is None else k_pad
    c = torch.empty((M, k_pad), device=a.device, dtype=torch.float32)
    _lib.check(lib.gif_linear_nn_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, a.stride(0), b.stride(0), k_pad, k_pad,
                                     float(scale), _stream()), "linear_nn")
    return c


def linear_tn(a, b, scale=1.0, n_valid=None, k_valid=None):
    """[n_valid, k_valid] = scale * a[M, :n_valid]^T @ b[M, :k_valid]."""
    lib = _lib.load()
    a, b = _mat(a, "linear_tn"), _mat(b, "linear_tn")
    M = a.shape[0]
    assert b.shape[0] == M
    N = a.shape[1] if n_valid is None else n_valid
    K = b.shape[1] if k_valid is None else k_valid
    c = torch.empty((N, K), device=a.device, dtype=torch.float32)
    _lib.check(lib.gif_linear_tn_f32(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, a.stride(0), b.stride(0), K, float(scale),
                                     _stream()), "linear_tn")
    return c


# ---- style path of ModulatedConv2d (csrc/linear.hip; include/gif_hip.h "Style path") ----
def weight_sq_sum(w):
    """wsq [Cout, Cin] = sum_taps w[Cout, Cin, kh, kw]^2, cached per parameter version like the packed weights."""
    lib = _lib.load()
    if not w.is_cuda or w.dtype != torch.float32 or w.dim() != 4:
        raise _lib.GifHipError("weight_sq_sum: need a 4-D fp32 device weight (no CPU fallback)")

    def build():
        wc = w.contiguous()
        O, I, KH, KW = wc.shape
        out = torch.empty((O, I), device=w.device, dtype=torch.float32)
        _lib.check(lib.gif_weight_sq_sum_f32(wc.data_ptr(), out.data_ptr(), O, I, KH * KW, _stream()), "weight_sq_sum")
        return out
    return _cached_weight_op(w, ("wsq",), build)


def style_demod(s, wsq, scale2, eps, cout_pad):
    """d [B, cout_pad] = rsqrt(scale2 * s[:, :Cin]^2 @ wsq^T + eps); padding columns 1."""
    lib = _lib.load()
    s, wsq = _mat(s, "style_demod"), _mat(wsq, "style_demod")
    B = s.shape[0]
    cout, cin = wsq.shape
    d = torch.empty((B, cout_pad), device=s.device, dtype=torch.float32)
    _lib.check(lib.gif_style_demod_f32(s.data_ptr(), wsq.data_ptr(), d.data_ptr(), B, cout, cin, s.stride(0), wsq.stride(0), cout_pad,
                                       cout_pad, float(scale2), float(eps), _stream()), "style_demod")
    return d


def style_demod_bwd_s(gd, d, wsq, s, gs_in, scale2):
    """gs_total [B, cin_pad] = gs_in + 2 * s * ((gd * (-scale2/2) * d^3)[:, :Cout] @ wsq); cin_pad = s.shape[1]."""
    lib = _lib.load()
    gd, d, wsq, s = _mat(gd, "style_demod_bwd_s"), _mat(d, "style_demod_bwd_s"), _mat(wsq, "style_demod_bwd_s"), _mat(s, "style_demod_bwd_s")
    B, cin_pad = s.shape
    cout, cin = wsq.shape
    if gd.shape != d.shape or gd.stride(0) != d.stride(0):
        gd, d = gd.contiguous(), d.contiguous()
    if gs_in is not None:
        gs_in = gs_in.contiguous()
        assert gs_in.shape == s.shape
    s = s.contiguous()
    out = torch.empty((B, cin_pad), device=s.device, dtype=torch.float32)
    _lib.check(lib.gif_style_demod_bwd_s_f32(gd.data_ptr(), d.data_ptr(), wsq.data_ptr(), s.data_ptr(), _p(gs_in), out.data_ptr(), B, cout,
                                             cin, gd.stride(0), wsq.stride(0), cin_pad, cin_pad, float(scale2), _stream()),
               "style_demod_bwd_s")
    return out


def style_demod_bwd_w(gd, d, s, cout, cin, scale2):
    """g_wsq [Cout, Cin] = (gd * (-scale2/2) * d^3)[:, :Cout]^T @ s[:, :Cin]^2."""
    lib = _lib.load()
    gd, d, s = _mat(gd, "style_demod_bwd_w"), _mat(d, "style_demod_bwd_w"), _mat(s, "style_demod_bwd_w")
    if gd.stride(0) != d.stride(0):
        gd, d = gd.contiguous(), d.contiguous()
    B = s.shape[0]
    out = torch.empty((cout, cin), device=s.device, dtype=torch.float32)
    _lib.check(lib.gif_style_demod_bwd_w_f32(gd.data_ptr(), d.data_ptr(), s.data_ptr(), out.data_ptr(), B, cout, cin, gd.stride(0),
                                             s.stride(0), float(scale2), _stream()), "style_demod_bwd_w")
    return out


def demod_wgrad(w, g_wsq):
    """gW [Cout, Cin, kh, kw] = 2 * w * g_wsq[:, :, None, None]."""
    lib = _lib.load()
    wc = w.contiguous()
    O, I, KH, KW = wc.shape
    g_wsq = g_wsq.contiguous()
    out = torch.empty_like(wc)
    _lib.check(lib.gif_demod_wgrad_f32(wc.data_ptr(), g_wsq.data_ptr(), out.data_ptr(), O, I, KH * KW, _stream()), "demod_wgrad")
    return out


for _name in ("linear_nt", "linear_nn", "linear_tn", "weight_sq_sum", "style_demod", "style_demod_bwd_s", "style_demod_bwd_w", "demod_wgrad"):
    globals()[_name] = _device_guard(globals()[_name])
del _name


FP32_MFMA_MODES = {"native": 0, "bf16x3": 1, "f16x2": 2}


def set_fp32_mfma_mode(mode):
    """How fp32 convolution contractions reach the matrix cores (process-wide, see include/gif_hip.h): "native" fp32 MFMA
    or "bf16x3" (exact three-way bf16 split of every fp32 operand, six bf16 MFMA products, fp32 accumulation).  Tensors
    are fp32 in HBM either way."""
    global _fp32_mode_cache
    _lib.check(_lib.load().gif_set_fp32_mfma_mode(FP32_MFMA_MODES[mode] if isinstance(mode, str) else int(mode)), "set_fp32_mfma_mode")
    _fp32_mode_cache = None


_fp32_mode_cache = None


def get_fp32_mfma_mode() -> str:
    global _fp32_mode_cache
    if _fp32_mode_cache is None:
        m = _lib.load().gif_get_fp32_mfma_mode()
        _fp32_mode_cache = {v: k for k, v in FP32_MFMA_MODES.items()}[m]
    return _fp32_mode_cache


def h2_fallback_stats(reset=False):
    """(guarded f16x2 launches that took the bf16x3 fallback,) on the current device since the last reset.  Synchronises."""
    out = (ctypes.c_uint64 * 2)()
    _lib.check(_lib.load().gif_h2_fallback_stats(out, 1 if reset else 0), "h2_fallback_stats")
    return int(out[0])


def prof_enable(on: bool):
    _lib.load().gif_prof_enable(1 if on else 0)


def prof_read(family: int):
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.load().gif_prof_read(family, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    return ms.value, fl.value, n.value
