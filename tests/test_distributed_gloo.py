"""N>1 path on CPU: world_size-2 gloo processes exercise FlatGradBucket (the one-all-reduce-per-optimiser-step
replacement of nn.DataParallel, train.py:344,356,358).  The G/D modules themselves are HIP-only, so a small torch
model stands in for them: what is checked is the data-parallel algebra — the mean of per-rank gradients equals the
gradient of the mean loss over the global batch, parameters with no gradient stay zero, replicas stay identical
after Adam steps."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gif_amd.train_step import FlatGradBucket, accumulate, flops_per_image


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LeakyReLU(0.2), torch.nn.Linear(16, 1))
    m.unused = torch.nn.Parameter(torch.ones(5))  # never receives a gradient (like G blocks above `step`)
    return m


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model()
    bucket = FlatGradBucket(m.parameters())
    opt = torch.optim.Adam(m.parameters(), lr=0.01, betas=(0.0, 0.99))
    g = torch.Generator().manual_seed(42)
    x_all = torch.randn(3, 8, 8, generator=g)  # 3 steps x global batch 8
    for step in range(3):
        x = x_all[step, rank * 4:(rank + 1) * 4]
        bucket.zero()
        torch.nn.functional.softplus(-m(x)).mean().backward()
        bucket.all_reduce_mean()
        if step == 0:
            first0 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
        opt.step()
    first = torch.cat([p.grad.reshape(-1) for p in m.parameters()])  # (taken after step 3; compare step-0 copy below)
    q.put((rank, first0.tolist(), torch.cat([p.detach().reshape(-1) for p in m.parameters()]).tolist()))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_global_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.tensor(a), torch.tensor(b)) for r, a, b in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the global batch
    m = _model()
    opt = torch.optim.Adam(m.parameters(), lr=0.01, betas=(0.0, 0.99))
    g = torch.Generator().manual_seed(42)
    x_all = torch.randn(3, 8, 8, generator=g)
    first = None
    for step in range(3):
        opt.zero_grad(set_to_none=False)
        for p in m.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        torch.nn.functional.softplus(-m(x_all[step])).mean().backward()
        if step == 0:
            first = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        opt.step()
    ref_params = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    for rank, flat, params in res:
        assert torch.allclose(flat, first, atol=1e-6), "mean of per-rank grads == global-batch grad"
        assert torch.allclose(params, ref_params, atol=1e-5), "replicas track the single-process run"
    assert torch.equal(res[0][2], res[1][2]), "replicas identical across ranks"
    assert (res[0][1][:5] == 0).all(), "parameter without gradient stays zero in the bucket"


def test_bucket_single_process_and_ema():
    m = _model()
    b = FlatGradBucket(m.parameters())
    assert b.flat.numel() >= sum(p.numel() for p in m.parameters())
    m(torch.ones(2, 8)).sum().backward()
    assert b.flat.abs().sum() > 0 and m.unused.grad.data_ptr() == b.flat.data_ptr()  # grads ARE the bucket
    b.all_reduce_mean()  # no process group: no-op
    m2 = _model()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    before = [p.clone() for p in m.parameters()]
    accumulate(m, m2, decay=0.75)
    for p, q0, q1 in zip(m.parameters(), before, m2.parameters()):
        assert torch.allclose(p, 0.75 * q0 + 0.25 * q1)
    assert abs(flops_per_image(256, 16) / 1e12 - 1.181) < 2e-3
