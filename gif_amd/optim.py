"""FlatAdam — torch.optim.Adam whose step() is ONE HIP launch over the flat gradient bucket (gif_adam_ema_step_f32),
optionally fused with the EMA update of the running generator (generic_utils.accumulate, my_utils/generic_utils.py:63-76).

Drop-in for the optimisers of train.py:364-381 in the one-process-per-GPU loop: same constructor hyper-parameters, same
update arithmetic (torch's single-tensor Adam, no weight decay / amsgrad), and a state_dict() in torch.optim.Adam's format
(`exp_avg`, `exp_avg_sq`, `step` per parameter) so checkpoints written by either load into the other (train.py:254-265).
Parameters keep their own storage; gradients / exp_avg / exp_avg_sq are views into flat buffers that share the bucket's
offset table.  Parameters outside the bucket (no gradient at the current resolution) are skipped exactly like torch's Adam
skips parameters whose .grad is None.  There is no CPU path: the kernel needs device tensors.
"""
import ctypes

import torch

from . import _lib


class _Chunk(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("ema", ctypes.c_void_p), ("flat_offset", ctypes.c_int64),
                ("n", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, bucket=None, ema_params=None):
        params = list(params)
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        if bucket is None:
            raise _lib.GifHipError("FlatAdam needs the FlatGradBucket that owns the parameters' gradients")
        self.bucket = bucket
        dev = bucket.flat.device
        if dev.type != "cuda":
            raise _lib.GifHipError("FlatAdam runs on the HIP device only (no CPU fallback); use torch.optim.Adam on CPU")
        self._m = torch.zeros_like(bucket.flat)
        self._v = torch.zeros_like(bucket.flat)
        self._step_t = torch.tensor(0.0)  # shared by every parameter's state (torch keeps one CPU scalar per parameter)
        # loss-scaled runs (found_inf given): the step count lives on the device and only advances on steps that were applied,
        # like torch's GradScaler-aware Adam; _step_t is refreshed from it when a state_dict is taken
        self._step_dev = None
        index = {id(p): i for i, p in enumerate(params)}
        self._ema = None
        if ema_params is not None:
            ema_params = list(ema_params)
            if len(ema_params) != len(params):
                raise _lib.GifHipError("FlatAdam: ema_params must mirror params one to one")
            self._ema = [ema_params[index[id(p)]] for p in bucket.params]
            active = set(id(p) for p in bucket.params)
            # parameters without a gradient still take part in the reference's EMA (accumulate walks all named parameters)
            self._ema_rest = [(ema_params[i], p) for i, p in enumerate(params) if id(p) not in active]
        for p, off in zip(bucket.params, bucket.offsets):
            n = p.numel()
            self.state[p] = {"step": self._step_t, "exp_avg": self._m[off:off + n].view_as(p),
                             "exp_avg_sq": self._v[off:off + n].view_as(p)}
        self._table = None
        self._table_key = None

    # ---- chunk table ---------------------------------------------------------------------------------------------
    def _chunks(self):
        key = tuple(p.data_ptr() for p in self.bucket.params)
        if self._ema is not None:
            key += tuple(e.data_ptr() for e in self._ema)
        if key == self._table_key:
            return self._table
        lib = _lib.load()
        cf = lib.gif_adam_chunk_floats()
        rows = []
        for k, (p, off) in enumerate(zip(self.bucket.params, self.bucket.offsets)):
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise _lib.GifHipError("FlatAdam: parameters must be contiguous fp32 tensors")
            e = self._ema[k] if self._ema is not None else None
            if e is not None and (not e.is_contiguous() or e.shape != p.shape or e.device != p.device):
                raise _lib.GifHipError("FlatAdam: EMA parameter does not mirror its parameter")
            n = p.numel()
            for s in range(0, n, cf):
                rows.append((p.data_ptr() + 4 * s, (e.data_ptr() + 4 * s) if e is not None else 0, off + s, min(cf, n - s), 0))
        arr = (_Chunk * len(rows))(*rows)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8) if rows else torch.empty(0, dtype=torch.uint8)
        self._table = (host.to(self.bucket.flat.device), len(rows))
        self._table_key = key
        return self._table

    # ---- step ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, ema_decay=None, inv_grad_scale=None, found_inf=None):
        """One Adam update of every parameter in the bucket; ema_decay (with ema_params given) also runs
        ema = ema*decay + (1-decay)*param in the same launch.  inv_grad_scale / found_inf: optional fp32 DEVICE scalars of a
        loss scaler — gradients are multiplied by the first, the update is skipped on device when the second is non-zero."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise _lib.GifHipError("FlatAdam supports one parameter group")
        g = self.param_groups[0]
        if g["weight_decay"] != 0 or g["amsgrad"] or g["maximize"]:
            raise _lib.GifHipError("FlatAdam implements plain Adam only (no weight decay / amsgrad / maximize)")
        self.bucket.attach()  # gradients must live in the bucket (re-aliases after zero_grad(set_to_none=True))
        b1, b2 = g["betas"]
        dev_step = None
        if found_inf is not None:
            if self._step_dev is None:
                self._step_dev = torch.full((), float(self._step_t.item()), device=self.bucket.flat.device)
            self._step_dev.add_(1.0 - found_inf.clamp(0.0, 1.0))  # a skipped step does not advance the bias corrections
            dev_step = self._step_dev
            t = 1  # (host-side corrections unused: the kernel derives them from the device counter)
        else:
            if self._step_dev is not None:  # leaving a loss-scaled phase: continue from the device count
                self._step_t.fill_(float(self._step_dev.item()))
                self._step_dev = None
            self._step_t += 1
            t = int(self._step_t.item())
        table, n = self._chunks()
        has_ema = self._ema is not None and ema_decay is not None
        lib = _lib.load()
        with torch.cuda.device(self.bucket.flat.device):  # launch on the parameters' device and its current stream
            _lib.check(lib.gif_adam_ema_step_f32(table.data_ptr(), n, self.bucket.flat.data_ptr(), self._m.data_ptr(),
                                                 self._v.data_ptr(), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                                 1.0 - b1 ** t, 1.0 - b2 ** t, float(ema_decay or 0.0), 1 if has_ema else 0,
                                                 None if inv_grad_scale is None else inv_grad_scale.data_ptr(),
                                                 None if found_inf is None else found_inf.data_ptr(),
                                                 None if dev_step is None else dev_step.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream), "adam_ema_step")
        # the kernel wrote through raw pointers: tell autograd the tensors changed (saved-tensor checks, caches keyed on it)
        torch.autograd.graph.increment_version(self.bucket.params)
        if has_ema:
            torch.autograd.graph.increment_version(self._ema)
            if self._ema_rest:
                e, p = [a for a, _ in self._ema_rest], [b for _, b in self._ema_rest]
                if found_inf is None:
                    torch._foreach_mul_(e, ema_decay)
                    torch._foreach_add_(e, p, alpha=1 - ema_decay)
                else:  # e += w * (p - e) with w = (1 - decay) on applied steps and 0 on skipped ones, decided on the device
                    w = (1.0 - ema_decay) * (1.0 - found_inf.clamp(0.0, 1.0))
                    torch._foreach_lerp_(e, p, [w] * len(e))
        return loss

    def zero_grad(self, set_to_none=True):
        """The gradients live in the flat bucket: clearing them means zeroing the bucket (a plain set-to-None would leave the
        previous step's values in the bucket for any parameter the next backward does not reach)."""
        self.bucket.zero()

    def state_dict(self):
        if self._step_dev is not None:
            self._step_t.fill_(float(self._step_dev.item()))
        return super().state_dict()

    # ---- checkpoint compatibility --------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam / FlatAdam state dicts: the loaded moments are copied into the flat buffers."""
        super().load_state_dict(state_dict)
        step = 0.0
        for p, off in zip(self.bucket.params, self.bucket.offsets):
            st = self.state.get(p)
            n = p.numel()
            m, v = self._m[off:off + n].view_as(p), self._v[off:off + n].view_as(p)
            if st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                step = max(step, float(st["step"]))
            else:
                m.zero_()
                v.zero_()
            self.state[p] = {"step": self._step_t, "exp_avg": m, "exp_avg_sq": v}
        self._step_t.fill_(step)
        self._step_dev = None  # a device-side count from before the load is stale: the next loss-scaled step re-seeds it from _step_t
