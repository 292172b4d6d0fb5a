"""One GIF training iteration on MI355X — the loop body of /root/reference/train.py:82-250 with
nn.DataParallel (train.py:344,356,358) replaced by ONE PROCESS PER GPU and a single RCCL all-reduce of the flat
gradient bucket per optimiser step (torch.distributed backend "nccl" == RCCL over xGMI).

Semantics kept from the reference: non-saturating logistic losses, R1 (weight 5.0) on the real images every
16th iteration, Adam(lr=0.002*r, betas=(0, 0.99**r)) with r = 16/17 (D) and 4/5 (G), EMA generator with decay
0.5**(32/10000) over named parameters, G frozen during the D step and vice versa, the optional generator
regularisers PATH_LEN_REG / DIRECT_GRAD_REG (train.py:203-215) and the embedding L2 term (:217-220).
Semantics that change with DataParallel -> per-rank replicas: parameters and buffers are broadcast
from rank 0 ONCE, at construction (DataParallel re-broadcasts them on every forward); minibatch-stddev groups are
formed inside the per-rank batch.
"""
import contextlib
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib
from . import losses
from .optim import FlatAdam


def requires_grad(model, flag=True):  # my_utils/generic_utils.py:58-60
    for p in model.parameters():
        p.requires_grad = flag


@torch.no_grad()
def accumulate(model1, model2, decay=0.999):  # my_utils/generic_utils.py:63-76 (EMA over named parameters)
    p1 = [p for _, p in model1.named_parameters()]
    p2 = [p for _, p in model2.named_parameters()]
    torch._foreach_mul_(p1, decay)
    torch._foreach_add_(p1, p2, alpha=1 - decay)


def _dist_on(group=None):
    """A process group with more than one rank exists.  GIF_FORCE_DIST=1 also takes the collective code paths in a group of
    ONE rank (tests: the RCCL calls — broadcast, asynchronous AVG all-reduce, waits — are exercised on a single-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("GIF_FORCE_DIST") == "1"


@torch.no_grad()
def broadcast_module_state(modules, src=0, group=None):
    """Make every rank's replica identical to rank `src`'s: all parameters AND buffers (the frozen ImgEmbedding code
    book is a buffer drawn with torch.randn per process).  What DDP does at construction; no-op without a process
    group of size > 1."""
    if not _dist_on(group):
        return
    via_host = dist.get_backend(group) != "nccl"  # gloo (tests): move device tensors through the host explicitly
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if via_host and t.is_cuda:
                h = t.data.cpu()
                dist.broadcast(h, src=src, group=group)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src=src, group=group)
    # writes through .data do not bump the tensors' version counters: packed / transformed weights cached from a forward that
    # ran before this broadcast would otherwise survive it on the non-source ranks
    from . import ops
    ops.clear_weight_cache()


class FlatGradBucket:
    """All ACTIVE parameter gradients of a model live in ONE contiguous fp32 buffer (p.grad are views into it), so the
    data-parallel exchange is exactly one all-reduce per optimiser step — D: 28.86 M floats = 115.5 MB, G at 256x256:
    31.24 M = 125.0 MB — instead of DataParallel's per-forward broadcast + per-backward reduce.

    `active`: optional predicate over the parameters; inactive ones (G blocks above the current resolution `step`,
    which receive no gradient) stay OUT of the bucket with p.grad = None, exactly like in the reference — Adam then
    skips them and creates no state for them — and they do not ride along in the all-reduce.

    The aliasing p.grad -> bucket is re-established by zero() and verified before every all-reduce: model.zero_grad()
    / optimizer.zero_grad(set_to_none=True) or a model.to() after construction would otherwise silently detach it."""

    ALIGN = 64  # floats: every view starts on a 256-byte boundary (vectorised optimiser / RCCL paths)

    def __init__(self, params, process_group=None, active=None):
        params = list(params)
        self.params = [p for p in params if active is None or active(p)]
        self.inactive = [p for p in params if not (active is None or active(p))]
        self.group = process_group
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        dev = self.params[0].device
        # one extra slot behind the last gradient rides along in the exchange: the loss scaler parks "an f16 gradient store
        # saturated on THIS rank" there as +inf, so the (summed / averaged) bucket is non-finite on EVERY rank and all of them
        # skip the step together — no second collective (DeviceLossScaler.end_backward).  The optimiser never touches it.
        self.flag_offset = n
        n += self.ALIGN
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flag_slot = self.flat[self.flag_offset:self.flag_offset + 1]
        self.views = [self.flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)]
        self._pending = None
        self._known_zero = True  # the whole buffer is zero (fresh / after zero()): attach() need not clear gradient-less views
        self.comm_events = None  # bench.py: a list collects (start, end) HIP-event pairs around every wait for an exchange
        self.attach()

    def attach(self):
        """(Re-)alias every active p.grad to its bucket view; gradients that live elsewhere are moved in first — all of them
        in ONE multi-tensor copy (after zero() that is every gradient of the backward pass: autograd hands a fresh tensor to a
        parameter whose .grad is None, where it would otherwise launch one accumulation add per parameter into the view)."""
        dst, src, stale = [], [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None or g.data_ptr() != v.data_ptr():
                if g is not None:
                    dst.append(v)
                    src.append(g.detach())
                else:
                    stale.append(v)  # no gradient this round: the view must not expose whatever the bucket held before
                p.grad = v
        with torch.no_grad():
            if dst:
                torch._foreach_copy_(dst, src)
            if stale and not self._known_zero:
                torch._foreach_zero_(stale)
        self._known_zero = False
        for p in self.inactive:
            p.grad = None

    def zero(self):
        """Zero the bucket and hand the parameters over to autograd with .grad = None; attach() (called by the exchange and by
        the optimiser) gathers what the backward pass produced."""
        self.wait()
        self.flat.zero_()
        self._known_zero = True
        for p in self.params:
            p.grad = None
        for p in self.inactive:
            p.grad = None

    def all_reduce_mean(self, async_op=False):
        """Average the bucket over the process group.  async_op=True only ENQUEUES the collective (it runs on the
        backend's own stream behind the work already queued on the current stream) — call wait() before the bucket is
        read; everything launched in between overlaps with the exchange."""
        self.attach()  # gather the backward pass's gradients (also without a process group: the bucket is what the optimiser reads)
        if not _dist_on(self.group):
            return
        ws = dist.get_world_size(self.group)
        if dist.get_backend(self.group) == "nccl":  # RCCL: the mean is taken inside the collective, no separate pass
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            self._pending = (work, None) if async_op else None
        elif self.flat.is_cuda:  # gloo with a device bucket (tests only): through the host, synchronously
            h = self.flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(h.div_(ws))
        else:  # gloo (CPU tests) has no AVG
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self._pending = (work, ws)
            else:
                self.flat.div_(ws)

    def wait(self):
        if self._pending is not None:
            work, div = self._pending
            self._pending = None
            ev = None
            if self.comm_events is not None and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            work.wait()
            if ev is not None:
                ev[1].record()
                self.comm_events.append(ev)
            if div:
                self.flat.div_(div)


class DeviceLossScaler:
    """Dynamic loss scaling for the f16-activation path, entirely on the device (no host sync per step): the loss is
    multiplied by `scale` before backward(); after the gradient exchange update() looks for non-finite values in the flat
    bucket, and FlatAdam.step(inv_grad_scale=, found_inf=) un-scales the gradients inside the Adam kernel or skips the update.
    An overflow halves the scale; `growth_interval` consecutive clean steps double it (torch.cuda.amp.GradScaler's policy).

    Saturated f16 gradient stores (they clamp at +-65504 and would never show up as inf in the fp32 weight gradients) raise a
    per-device flag word while a watch window is open.  Protocol of one optimiser step of one network (round 4, advisor
    findings): begin_step() clears the flag; every GRADIENT pass of the step — the regularisers' inner autograd.grad calls
    and the final backward() — runs inside `with scaler.watching():`; end_backward(bucket) snapshots the flag into this
    scaler's own device scalar right after backward() (so launches of the OTHER network issued before a deferred update
    cannot leak into it) and writes it as +inf into the bucket's flag slot, which rides along in the gradient exchange: every
    rank then sees a non-finite bucket and all replicas skip the step and halve their scale together."""

    def __init__(self, device, init_scale=2.0 ** 12, growth_interval=1000, max_scale=2.0 ** 20):
        self.scale = torch.tensor(float(init_scale), device=device)
        self.inv_scale = torch.tensor(1.0 / float(init_scale), device=device)
        self.found_inf = torch.zeros((), device=device)
        self.sat = torch.zeros((), device=device)  # this rank's "a gradient store saturated" snapshot of the current step
        self._snapshotted = False
        self._good = torch.zeros((), device=device)
        self.growth_interval, self.max_scale = float(growth_interval), float(max_scale)
        self.skipped = torch.zeros((), device=device)  # number of skipped steps (read it after a synchronize)

    def _hip(self):
        return self.scale.is_cuda

    def begin_step(self):
        """Once per optimiser step, before its first gradient pass: clear the device's flag word (the window stays closed)."""
        self.sat.zero_()
        self._snapshotted = False
        if self._hip():
            from . import _lib
            with torch.cuda.device(self.scale.device):
                lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
                _lib.check(lib.gif_f16_overflow_clear(st), "f16_overflow_clear")
                _lib.check(lib.gif_f16_overflow_watch(0), "f16_overflow_watch")

    @contextlib.contextmanager
    def watching(self):
        """f16 launches issued inside the block check their stores (forward passes outside it pay nothing)."""
        if self._hip():
            from . import _lib
            _lib.check(_lib.load().gif_f16_overflow_watch(1), "f16_overflow_watch")
        try:
            yield
        finally:
            if self._hip():
                from . import _lib
                _lib.check(_lib.load().gif_f16_overflow_watch(0), "f16_overflow_watch")

    def begin_backward(self):
        """Short form for a step whose only gradient pass is backward(): begin_step() + open the window; end_backward() or
        update() closes it."""
        self.begin_step()
        if self._hip():
            from . import _lib
            _lib.check(_lib.load().gif_f16_overflow_watch(1), "f16_overflow_watch")

    def _snapshot(self):
        """sat |= the device's flag word (closes the window)."""
        if self._hip():
            from . import _lib
            with torch.cuda.device(self.scale.device):
                _lib.check(_lib.load().gif_f16_overflow_or_into(self.sat.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           "f16_overflow_or_into")

    @torch.no_grad()
    def end_backward(self, bucket=None):
        """Right after backward(), BEFORE the gradient exchange is enqueued: snapshot the flag; with a bucket, park it in the
        bucket's flag slot as +inf (0 otherwise) so that the exchange spreads it to every rank."""
        self._snapshot()
        self._snapshotted = True
        if bucket is not None:
            bucket.flag_slot.copy_(torch.where(self.sat > 0, float("inf"), 0.0).reshape(1))

    @torch.no_grad()
    def update(self, flat):
        """Call once per optimiser step BEFORE the Adam launch, with the (already exchanged) gradient bucket.  found_inf is a
        function of the exchanged bucket alone (incl. its flag slot), i.e. identical on every rank; a caller that never called
        end_backward() (single-process use without a bucket) still gets the local flag OR-ed in here — but after
        end_backward() the device's flag word belongs to whoever cleared it next (the other network's step)."""
        if not self._snapshotted:
            self._snapshot()
        self._snapshotted = False
        self.found_inf.copy_(torch.maximum((~torch.isfinite(flat).all()).to(torch.float32), (self.sat > 0).to(torch.float32)))
        bad = self.found_inf.clone()
        self.inv_scale.copy_(1.0 / self.scale)  # the scale the gradients in `flat` were produced with
        self.skipped.add_(bad)
        good = (self._good + 1.0) * (1.0 - bad)
        grow = (good >= self.growth_interval).to(torch.float32)
        self._good.copy_(good * (1.0 - grow))
        new_scale = torch.where(bad > 0, self.scale * 0.5, torch.where(grow > 0, self.scale * 2.0, self.scale))
        self.scale.copy_(new_scale.clamp(1.0, self.max_scale))


def _g_param_active(generator, step):
    """Predicate: does this generator parameter receive a gradient at resolution `step`?  Blocks / ToRGB layers above
    `step` are never executed (Generator.forward breaks at i == step, stg2_generator.py:205-206)."""
    dead = set()
    gen = getattr(generator, "generator", None)
    if gen is not None:
        for lst in (gen.progression, gen.to_rgb):
            for i, m in enumerate(lst):
                if i > step:
                    dead.update(id(p) for p in m.parameters())
    return lambda p: id(p) not in dead


class GifTrainer:
    """Holds G, D, the EMA generator, both Adam optimisers and the two gradient buckets; step() is one iteration.

    Multi-GPU (one process per GPU, torch.distributed initialised by the caller): construction broadcasts rank 0's
    G / D / G_ema parameters and buffers to every rank; each optimiser step all-reduces one flat bucket.  With
    overlap_comm (default when a process group of size > 1 exists) each network's all-reduce + Adam update are deferred to
    the point where that network is next needed — D's until the D forward of the G step (the exchange runs under the
    generator forward), G's until the generator forward of the next iteration (under the D forward on the real images) —
    so neither exchange sits on the critical path; flush() completes anything pending (call it before reading weights or
    the EMA generator, e.g. to checkpoint)."""

    def __init__(self, generator, discriminator, g_running, step=6, alpha=1.0, r1_every=16, gen_reg_type='None',
                 embedding_reg_weight=0.0, lr=0.002, fused_adam=None, process_group=None,
                 reuse_generator_forward=False, overlap_comm=None, sync_initial_state=True, act_dtype=None,
                 loss_scale=2.0 ** 12, fuse_d_passes=None, texture_loss=None, adaptive_interp_loss=False, max_ids=None,
                 flame_un_normalizer=None):
        self.G, self.D, self.G_ema = generator, discriminator, g_running
        # act_dtype=torch.float16: BASELINE config 5 — f16 activations in G and D (fp32 master weights, demodulation,
        # accumulation, optimiser), dynamic loss scaling on the device.  None keeps whatever the modules are set to.
        if act_dtype is not None:
            for m in (generator, discriminator, g_running):
                m.set_activation_dtype(act_dtype)
        self.f16 = getattr(generator, "act_dtype", torch.float32) == torch.float16 or \
            getattr(discriminator, "act_dtype", torch.float32) == torch.float16
        self.res_step, self.alpha, self.r1_every = step, alpha, r1_every
        self.gen_reg_type = gen_reg_type.upper()
        self.embedding_reg_weight = embedding_reg_weight
        self.group = process_group
        if sync_initial_state:
            broadcast_module_state((generator, discriminator, g_running), 0, process_group)
        g_ratio, d_ratio = 4 / (4 + 1), 16 / (16 + 1)  # train.py:364-381
        on_gpu = next(generator.parameters()).is_cuda
        self.g_bucket = FlatGradBucket(generator.parameters(), process_group, _g_param_active(generator, step))
        self.d_bucket = FlatGradBucket(discriminator.parameters(), process_group)
        hip_adam = on_gpu if fused_adam is None else fused_adam
        if hip_adam:  # one HIP launch per optimiser step over the flat buffers (+ the EMA of the generator)
            self.g_optim = FlatAdam(generator.parameters(), lr=lr * g_ratio, betas=(0.0, 0.99 ** g_ratio),
                                    bucket=self.g_bucket, ema_params=[p for _, p in g_running.named_parameters()])
            self.d_optim = FlatAdam(discriminator.parameters(), lr=lr * d_ratio, betas=(0.0, 0.99 ** d_ratio),
                                    bucket=self.d_bucket)
        else:
            self.g_optim = torch.optim.Adam(generator.parameters(), lr=lr * g_ratio, betas=(0.0, 0.99 ** g_ratio))
            self.d_optim = torch.optim.Adam(discriminator.parameters(), lr=lr * d_ratio, betas=(0.0, 0.99 ** d_ratio))
        self.g_scaler = self.d_scaler = None
        if self.f16:
            if not hip_adam:
                raise losses.ops._lib.GifHipError("f16 activations need the HIP optimiser (loss scaling is fused into FlatAdam)")
            dev = next(generator.parameters()).device
            self.g_scaler, self.d_scaler = DeviceLossScaler(dev, loss_scale), DeviceLossScaler(dev, loss_scale)
        self.pl_reg = losses.PathLengthRegularizor() if self.gen_reg_type == 'PATH_LEN_REG' else None
        # Texture-space interpolation loss of run 29 (train.py:222-238, configurations.py:217): texture_loss is a
        # losses.InterpolatedTextureLoss whose FLAME-dependent parts (texture decoder, condition renderer) were injected;
        # g_step(flame_batch=) applies it on every generator step, like `args.apply_texture_space_interpolation_loss`.
        # adaptive_interp_loss: train.py:233-234; max_ids: args.embedding_vocab_size (default: the generator's code book size);
        # flame_un_normalizer: dataset.un_normalize_flame (train.py:228), identity when None.
        self.texture_loss = texture_loss
        self.adaptive_interp_loss = bool(adaptive_interp_loss)
        self.max_ids = max_ids if max_ids is not None else getattr(generator, "embedding_vocab_size", 1)
        self.flame_un_normalizer = flame_un_normalizer
        # Optional (off by default, NOT used by bench.py): the reference runs the generator twice per iteration on
        # identical inputs and identical weights (train.py:157 and :197 — G only changes at :243).  With this flag the
        # forward is executed once with autograd enabled; the D step consumes fake.detach(), the G step back-propagates
        # through the same graph.  Bit-identical losses and updates, one generator forward (9 % of the FLOPs) less.
        self.reuse_generator_forward = reuse_generator_forward
        # D step: train.py:142 and :169 call the discriminator twice, on the real and on the generated batch.  The two calls share
        # the weights and are independent per sample (the minibatch-stddev groups are formed inside each call), so they run as ONE
        # pass over [real; fake] with the statistic taken per half (Discriminator.stddev_chunks): identical scores, the same
        # FLOPs, half the launches of the D step, 2 x the rows for the 4x4 .. 16x16 layers that cannot fill the chip at batch 32,
        # and every D parameter receives ONE gradient instead of two that autograd has to add.  R1 iterations keep the separate
        # calls (the penalty differentiates the real half only).  GIF_FUSE_D=0 / fuse_d_passes=False: A/B.
        self.overlap_comm = _dist_on(process_group) if overlap_comm is None else overlap_comm
        # The fused pass is the default at EVERY rank count (round 6): a multi-GPU run then issues exactly the launch sequence every
        # single-GPU number was taken on, and `comm_exposed_ms` is the only difference between N = 1 and N > 1.  Its price with deferred
        # exchanges: the fused pass needs the generated batch first, so G's exchange + Adam + EMA complete before any D work starts
        # and G's all-reduce (125 MB) is exposed — predicted 0.7-1.5 ms per step on 8 GPUs — where the two-call path would
        # hide it under D's forward on the real images at ~1 ms of extra launches.  fuse_d_passes=False (bench.py --two-call-d) keeps
        # the two-call schedule for the A/B on hardware; GIF_FUSE_D=0 does the same under GIF_EXPERIMENTAL=1.
        fuse_default = _lib.knob("GIF_FUSE_D", "1") != "0"
        self.fuse_d_passes = fuse_default if fuse_d_passes is None else bool(fuse_d_passes)
        # Experiment (review item 2 of round 4, profiles/r5_two_streams.txt): in the two-call D step the no-grad generator forward that
        # produces the fake batch is independent of D's forward on the real images; GIF_TWO_STREAMS=1 (+ GIF_EXPERIMENTAL=1) runs it on
        # a second HIP stream.  Packed-weight cache entries built on the side stream would later be consumed (and freed) on the main
        # stream — a cross-stream use-after-free hazard in the caching allocator (advisor, round 5) — so the cache is switched off
        # for the side-stream forward (it re-packs its weights; `fake`, the only tensor that leaves the block, gets record_stream).
        self.two_streams = _lib.knob("GIF_TWO_STREAMS", "0") == "1"
        self._side = None
        self._d_update_pending = False
        self._g_update_pending = False
        self.g_running_decay = 0.5 ** (32 / (10 * 1000))
        self.G_ema.train(False)
        requires_grad(self.G, False)  # train.py:68
        requires_grad(self.D, True)

    # ---- optimiser updates -------------------------------------------------------------------------------------
    def _d_optim_step(self):
        if self.d_scaler is not None:
            self.d_scaler.update(self.d_bucket.flat)
            self.d_optim.step(inv_grad_scale=self.d_scaler.inv_scale, found_inf=self.d_scaler.found_inf)
        else:
            self.d_optim.step()

    def _finish_d_update(self):
        if self._d_update_pending:
            self._d_update_pending = False
            self.d_bucket.wait()
            self._d_optim_step()

    def flush(self):
        """Complete the deferred updates (overlap_comm)."""
        self._finish_d_update()
        self._finish_g_update()

    def ema_generator(self):
        """The running-average generator with every deferred update applied (overlap_comm defers G's exchange + Adam + EMA to the
        next use of G): use this — or call flush() — before sampling from G_ema / reading weights mid-training."""
        self.flush()
        return self.G_ema

    def _finish_g_update(self):
        if self._g_update_pending:
            self._g_update_pending = False
            self.g_bucket.wait()
            self._g_optim_step()

    def _g_update(self):
        if self.overlap_comm:
            self.g_bucket.all_reduce_mean(async_op=True)
            self._g_update_pending = True  # completed right before G is used again
            return
        self.g_bucket.all_reduce_mean()
        self._g_optim_step()

    def _g_optim_step(self):
        if self.g_scaler is not None:
            self.g_scaler.update(self.g_bucket.flat)
            self.g_optim.step(ema_decay=self.g_running_decay, inv_grad_scale=self.g_scaler.inv_scale,
                              found_inf=self.g_scaler.found_inf)
        elif isinstance(self.g_optim, FlatAdam):
            self.g_optim.step(ema_decay=self.g_running_decay)  # Adam + generic_utils.accumulate in one launch
        else:
            self.g_optim.step()
            accumulate(self.G_ema, self.G, self.g_running_decay)

    # ---- the two halves of an iteration ------------------------------------------------------------------------
    def d_step(self, i, real_image, cond, input_indices, fake=None):
        """train.py:82-178"""
        G, D = self.G, self.D
        self._finish_d_update()
        requires_grad(D, True)
        self.d_bucket.zero()
        r1_step = bool(self.r1_every) and (i + 1) % self.r1_every == 0
        sc = self.d_scaler
        watch = sc.watching if sc is not None else contextlib.nullcontext
        if sc is not None:
            sc.begin_step()
        # the reference marks the real image as requiring grad on every iteration (train.py:135-136) but only uses the
        # gradient on R1 iterations; requesting it only then skips a dead dgrad of D's first layer otherwise
        real_image = real_image.detach().requires_grad_(r1_step)
        if self.fuse_d_passes and not r1_step and hasattr(D, "stddev_chunks"):
            if fake is None:
                self._finish_g_update()
                with torch.no_grad():
                    fake = G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)[0]
            nb = real_image.shape[0]
            D.stddev_chunks = 2
            try:
                scores, _ = D([torch.cat((real_image, fake.detach().to(real_image.dtype)), dim=0)], condition=torch.cat((cond, cond), dim=0),
                              step=self.res_step, alpha=self.alpha)
            finally:
                D.stddev_chunks = 1
            d_loss = F.softplus(-scores[:nb]).mean() + F.softplus(scores[nb:]).mean()
            return self._d_backward_and_update(d_loss, sc, watch)
        side_fake = False
        if fake is None and self.two_streams:
            self._finish_g_update()
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(main)  # the inputs and G's parameters are final on the main stream
            from . import ops
            cache_was, ops.WEIGHT_CACHE = ops.WEIGHT_CACHE, False  # nothing allocated on the side stream may outlive the block
            try:
                with torch.cuda.stream(self._side), torch.no_grad():
                    fake = G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)[0]
            finally:
                ops.WEIGHT_CACHE = cache_was
            side_fake = True
        real_scores, _ = D([real_image], condition=cond, step=self.res_step, alpha=self.alpha)
        real_loss = F.softplus(-real_scores).mean()
        if r1_step:
            # f16: the inner gradient (d scores / d image) is taken on scores pre-multiplied by a constant so that the
            # activation gradients of the first backward stay inside the f16 range; the result is divided back
            with watch():  # the inner gradient's f16 stores are gradient stores too (a clamped one falsifies the penalty)
                real_loss = real_loss + losses.grad_penalty_loss([real_image], real_scores, step=None,
                                                                 grad_scale=(2.0 ** 10 if self.f16 else 1.0)).mean()
        if fake is None:
            self._finish_g_update()  # G's exchange of the previous iteration ran under the D forward above
            with torch.no_grad():  # the reference detaches the fake image right after the forward (train.py:160)
                fake = G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)[0]
        if side_fake:
            torch.cuda.current_stream().wait_stream(self._side)
            fake.record_stream(torch.cuda.current_stream())
        fake_scores, _ = D([fake.detach()], condition=cond, step=self.res_step, alpha=self.alpha)
        fake_loss = F.softplus(fake_scores).mean()
        d_loss = real_loss + fake_loss
        return self._d_backward_and_update(d_loss, sc, watch)

    def _d_backward_and_update(self, d_loss, sc, watch):
        with watch():
            (d_loss if sc is None else d_loss * sc.scale).backward()
        if sc is not None:
            sc.end_backward(self.d_bucket)  # before the exchange: this rank's saturation flag travels in the bucket
        if self.overlap_comm:
            self.d_bucket.all_reduce_mean(async_op=True)
            self._d_update_pending = True  # completed right before D is used again
        else:
            self.d_bucket.all_reduce_mean()
            self._d_optim_step()
        return d_loss.detach()

    def g_step(self, cond, input_indices, fake=None, flame_batch=None):
        """train.py:189-252.  flame_batch: the FLAME labels `flm_lbls` [B, >=159] of the batch — with a texture_loss installed the
        texture-space interpolation loss (train.py:222-238) is added on them."""
        G, D = self.G, self.D
        self._finish_g_update()
        requires_grad(G, True)
        requires_grad(D, False)
        self.g_bucket.zero()
        direct_reg = self.gen_reg_type == 'DIRECT_GRAD_REG'
        sc = self.g_scaler
        watch = sc.watching if sc is not None else contextlib.nullcontext
        if sc is not None:
            sc.begin_step()
        if direct_reg:
            cond = cond.detach().requires_grad_(True)
            fake = None  # the regulariser differentiates the images w.r.t. THIS condition tensor
        if fake is None:
            fake = G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)
        self._finish_d_update()  # the exchange of D's gradients ran under the generator forward above
        pred, _ = D(fake, condition=cond.detach(), step=self.res_step, alpha=self.alpha)
        loss = F.softplus(-pred).mean()
        if self.pl_reg is not None:
            with watch():  # (covers the regulariser's own generator forward as well: its stores are checked, harmlessly)
                loss = loss + 2 * self.pl_reg.path_length_reg(G, step=self.res_step, alpha=self.alpha,
                                                              input_indices=input_indices, cond=cond)
        elif direct_reg:
            # train.py:209-215: changes of the condition should change the image as little as possible.  The reference
            # adds the per-sample [B] penalty to the scalar loss and calls .backward() on the result, which only works
            # for batch 1; here the penalty is averaged over the batch (identical for batch 1).
            with watch():
                loss = loss + 1e-8 * 8 * losses.grad_penalty_loss([cond], torch.pow(fake[-1], 2), step=None).mean()
        if self.embedding_reg_weight:
            loss = loss + self.embedding_reg_weight * losses.l2_reg(G.z_to_w)
        if self.texture_loss is not None and flame_batch is not None:
            # train.py:222-238: the texture must stay the same when the face moves with different FLAME parameters
            flm_intrp_batch = losses.interpolate_flame_labels(flame_batch)
            if self.flame_un_normalizer is not None:
                flm_intrp_batch = self.flame_un_normalizer(flm_intrp_batch)
            interp_loss = self.texture_loss.tex_sp_intrp_loss(flm_intrp_batch, G, step=self.res_step, alpha=self.alpha,
                                                              max_ids=self.max_ids)
            if self.adaptive_interp_loss:
                interp_loss = interp_loss * (0.25 * loss.detach() / interp_loss.detach())
            loss = loss + interp_loss
        with watch():
            (loss if sc is None else loss * sc.scale).backward()
        if sc is not None:
            sc.end_backward(self.g_bucket)
        self._g_update()
        requires_grad(G, False)
        return loss.detach()

    def step(self, i, real_image, cond, input_indices, flame_batch=None):
        if self.reuse_generator_forward:
            self._finish_g_update()
            requires_grad(self.G, True)
            fake = self.G(cond, None, step=self.res_step, alpha=self.alpha, input_indices=input_indices)
            d_loss = self.d_step(i, real_image, cond, input_indices, fake=fake[0])
            g_loss = self.g_step(cond, input_indices, fake=fake, flame_batch=flame_batch)
            return d_loss, g_loss
        d_loss = self.d_step(i, real_image, cond, input_indices)
        g_loss = self.g_step(cond, input_indices, flame_batch=flame_batch)
        return d_loss, g_loss


# Contraction-only MACs per image at 256x256 measured on the reference: used for the roofline line
F_G_256, F_D_256 = 52.33e9, 46.58e9


def flops_per_image(res=256, r1_every=16, generator_forwards=2, extra_generator_fwd_bwd=0.0):
    """generator_forwards: 2 = the reference's iteration (train.py:155, :195); 1 = GifTrainer(reuse_generator_forward=True);
    extra_generator_fwd_bwd: additional generator forward + backward passes per image of the batch (the texture-interpolation loss
    runs one on batch - 1 images: (B - 1) / B)."""
    table = {64: (17.34e9, 16.46e9), 128: (33.76e9, 31.51e9), 256: (F_G_256, F_D_256), 512: (75.82e9, 61.69e9),
             1024: (111.71e9, 76.87e9)}
    fg, fd = table[res]
    fl = 2 * ((2 + generator_forwards + 3 * extra_generator_fwd_bwd) * fg + 8 * fd)
    if r1_every:
        fl += 2 * 3 * fd / r1_every
    return fl
