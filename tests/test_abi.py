"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/gif_hip.h
declares, the ctypes prototypes cover all of them, argument validation works without a GPU, and the product
path refuses CPU tensors instead of silently falling back."""
import ctypes
import os
import re

import pytest
import torch

from gif_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gif_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gif_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gif_hip.h but not exported by libgif_hip.so"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.gif_abi_version() == 4  # 4: the f16x2 contraction mode (round 5)


def test_argument_validation_without_gpu():
    lib = _lib.load()
    assert lib.gif_rasterize_workspace_bytes(2, 100, 65, 8) == (2 * 2 + 2 * 2 * 100) * 4  # per (image, 64x64 tile): counter + F-entry list
    rp, cp = ctypes.c_int(), ctypes.c_int()
    assert lib.gif_conv2d_pack_dims(512, 512, ctypes.byref(rp), ctypes.byref(cp)) == 0
    assert (rp.value, cp.value) == (512, 512)
    assert lib.gif_conv2d_pack_dims(3, 24, ctypes.byref(rp), ctypes.byref(cp)) == 0
    assert rp.value % 32 == 0 and cp.value % 8 == 0 and rp.value >= 3 and cp.value >= 24
    # bad geometry is rejected before any launch
    g = _lib.ConvGeom(1, 8, 8, 6, 8, 8, 8, 3, 3, 1, 1)  # Cb = 6 is not a multiple of 4
    e = _lib.ConvEpilogue()
    rc = lib.gif_conv2d_fwd_f32(None, None, None, ctypes.byref(g), ctypes.byref(e), None)
    assert rc == -1 and b"multiples of 4" in lib.gif_last_error()
    assert lib.gif_upfirdn2d_f32(None, None, None, *([1] * 13), ctypes.byref(e), None) == -1
    # empty work is a no-op, not an error (F = 0 faces)
    assert lib.gif_rasterize_f32(None, None, None, None, 2, 0, 4, 4, None, None) == 0


def test_fp32_mfma_mode_api_without_gpu():
    """include/gif_hip.h: process-wide numerics mode of the fp32 contractions; default f16x2 unless GIF_FP32_MFMA says otherwise."""
    import os
    from gif_amd import ops
    lib = _lib.load()
    before = lib.gif_get_fp32_mfma_mode()
    if os.environ.get("GIF_FP32_MFMA") in (None, "bf16x3", "f16x2"):
        assert before in (0, 1, 2)  # (an earlier test of this process may have switched it)
    try:
        assert lib.gif_set_fp32_mfma_mode(0) == 0 and lib.gif_get_fp32_mfma_mode() == 0
        ops.set_fp32_mfma_mode("bf16x3")
        assert ops.get_fp32_mfma_mode() == "bf16x3" and lib.gif_get_fp32_mfma_mode() == 1
        ops.set_fp32_mfma_mode("f16x2")
        assert ops.get_fp32_mfma_mode() == "f16x2" and lib.gif_get_fp32_mfma_mode() == 2 and ops.split_mode()
        assert lib.gif_set_fp32_mfma_mode(5) == -1 and b"unknown mode" in lib.gif_last_error()
        assert lib.gif_pack_weight_f32h2_bytes(3, 3, 128, 128) == 2 * 128 * 4 + 9 * 2 * 128 * 128 * 2
    finally:
        lib.gif_set_fp32_mfma_mode(before)
        ops._fp32_mode_cache = None
    rp, cp = ctypes.c_int(), ctypes.c_int()
    assert lib.gif_conv2d_pack_dims_x3(128, 24, ctypes.byref(rp), ctypes.byref(cp)) == 0 and (rp.value, cp.value) == (128, 32)
    assert lib.gif_winograd_pack_dims_x3(192, 100, ctypes.byref(rp), ctypes.byref(cp)) == 0 and (rp.value, cp.value) == (256, 128)
    assert lib.gif_conv2d_x3_eligible(64, 24) == 1 and lib.gif_conv2d_x3_eligible(64, 20) == 0
    # tap-dense K order: 32-float steps for 9 taps of cin_act channels, 0 where the mode does not apply
    steps = lib.gif_conv2d_x3_tapdense_steps
    assert [steps(c, 3, 3) for c in (8, 12, 24, 28)] == [3, 4, 7, 8]
    assert steps(32, 3, 3) == 0 and steps(4, 3, 3) == 0 and steps(24, 1, 1) == 0 and steps(10, 3, 3) == 0
    # (the dispatch itself is narrower: measured per shape, ops.x3_tapdense)
    spec = ops.ConvSpec(3, 3, 1, 1)
    ops._fp32_mode_cache = None
    if ops.split_mode():  # (bf16x3, and f16x2: its thin layers keep the tap-dense bf16x3 kernels)
        import torch
        assert ops.x3_tapdense(torch.float32, 24, spec, False, {}, 128) and ops.x3_tapdense(torch.float32, 12, spec, True, {}, 24)
        assert not ops.x3_tapdense(torch.float32, 24, spec, True, {}, 12) and not ops.x3_tapdense(torch.float32, 8, spec, False, {}, 12)
        assert not ops.x3_tapdense(torch.float32, 24, ops.ConvSpec(3, 3, 2, 0), True, {}, 128)
        assert not ops.x3_tapdense(torch.float32, 24, spec, False, {"in_scale": object()}, 128)


def test_no_cpu_fallback():
    from gif_amd import functional as GF, ops
    with pytest.raises(_lib.GifHipError):
        ops.nhwc(torch.zeros(1, 4, 4, 4))
    with pytest.raises(_lib.GifHipError):
        GF.bias_act(torch.zeros(1, 4, 4, 4), None)
    from gif_amd import standard_rasterize as sr
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        sr.standard_rasterize(torch.zeros(1, 1, 3, 3), torch.zeros(1, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int32),
                              torch.zeros(1, 4, 4, 3), 4, 4)


def test_state_dict_keys_match_reference_template():
    """Reference state_dicts must load strict=True: key names and shapes are pinned by the golden template."""
    import contextlib, io
    from gif_amd.generator import StyledGenerator
    from gif_amd.discriminator import Discriminator
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "stylegan2_golden.pt"), weights_only=True)
    with contextlib.redirect_stdout(io.StringIO()):
        g = StyledGenerator(embedding_vocab_size=50, rendered_flame_ascondition=True, normal_maps_as_cond=True)
        d = Discriminator(size=32, num_color_chnls=9)
    assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == gold["g_template"]
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == gold["d_template"]
    for k, v in gold["g_kernels"].items():
        assert torch.equal(g.state_dict()[k], v), k
    for k, v in gold["d_kernels"].items():
        assert torch.equal(d.state_dict()[k], v), k


def test_lib_module_loads_torch_before_the_hip_library():
    """gif_amd._lib must import torch BEFORE dlopen-ing libgif_hip.so: torch ships its own libamdhip64, and a process that binds the
    ROCm install's copy first (library loaded before torch) ends up with two HIP runtimes — every launch then fails with "no
    ROCm-capable device is detected" (round 6: build() + smoke() in one process).  Checked in a fresh interpreter."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); from gif_amd import _lib; assert 'torch' in sys.modules; _lib.load(); print('ok')" % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_experimental_gate_over_ab_knobs(monkeypatch):
    """A/B / ablation knobs are honoured only together with GIF_EXPERIMENTAL=1 (round 6): gif_amd._lib.knob returns the default
    otherwise; the documented switches (GIF_FP32_MFMA) are not routed through it."""
    from gif_amd import _lib
    monkeypatch.setenv("GIF_WINOGRAD", "0")
    monkeypatch.delenv("GIF_EXPERIMENTAL", raising=False)
    assert _lib.knob("GIF_WINOGRAD", "1") == "1"
    monkeypatch.setenv("GIF_EXPERIMENTAL", "0")
    assert _lib.knob("GIF_WINOGRAD", "1") == "1"
    monkeypatch.setenv("GIF_EXPERIMENTAL", "1")
    assert _lib.knob("GIF_WINOGRAD", "1") == "0"
    assert _lib.knob("GIF_NOT_SET_ANYWHERE", "7") == "7"


def test_winograd_channel_rule_follows_the_contraction_mode():
    """Round 6 dispatch: in the f16x2 mode the Winograd route starts at 256 channels on the thinner side (forward / data gradient and
    weight gradient alike); bf16x3 and native keep the tile-count rule alone; an explicit override wins."""
    from gif_amd import ops
    spec = ops.ConvSpec(3, 3, 1, 1)
    before = ops.get_fp32_mfma_mode()
    try:
        ops.set_fp32_mfma_mode("f16x2")
        assert not ops.winograd_eligible(spec, 32, 256, 256, 128, 128)
        assert ops.winograd_eligible(spec, 32, 128, 128, 256, 256)
        assert not ops.winograd_eligible(spec, 32, 128, 128, 256, 128)  # the thinner side counts
        assert ops.winograd_eligible(spec, 32, 256, 256, 128, 128, min_c=0)
        for mode in ("bf16x3", "native"):
            ops.set_fp32_mfma_mode(mode)
            assert ops.winograd_eligible(spec, 32, 256, 256, 128, 128)
    finally:
        ops.set_fp32_mfma_mode(before)
