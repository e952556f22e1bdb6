"""The data-parallel fine-tune step (multilingual_kws_amd/parallel.py) on CPU: world_size 2, gloo.
The device head is stood in for by an oracle-backed object with the same three methods, so what is
tested is the collective logic: one all-reduce(sum) of the flat gradient, grad_scale = 1/world,
identical parameters on every rank, equality with single-process training on the merged batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multilingual_kws_amd import parallel
from oracle import head_oracle as ho

DIMS = (32, 6, 3)


class OracleHead:
    """head-like: loss_grad / grad_view / adam_step backed by oracle/head_oracle.py (float64)."""

    def __init__(self, p0):
        self.p = np.asarray(p0, dtype=np.float64).copy()
        self.opt = None
        self.buf = torch.zeros(len(p0) + 2, dtype=torch.float64)     # gradients | sum of row losses | #correct
        self.g = self.buf[:-2]
        self.collectives = 0

    def loss_grad(self, emb, labels):
        _, g, ncorrect, lsum = ho.loss_and_grad(self.p, emb.numpy(), labels.numpy(), *DIMS)
        self.g.copy_(torch.from_numpy(g))
        self.buf[-2], self.buf[-1] = lsum, float(ncorrect)
        return self.buf[-2:].clone()

    def grad_view(self, with_stats=False):
        return self.buf if with_stats else self.g

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        if self.opt is None:
            self.opt = ho.KerasAdam(len(self.p), lr, beta1, beta2, eps)
        self.p = self.opt.step(self.p, self.g.numpy() * grad_scale)


def _data(seed, n):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((n, DIMS[0]))), torch.from_numpy(rng.integers(0, 3, n))


def _p0():
    n = DIMS[0] * DIMS[1] + DIMS[1] + DIMS[1] * DIMS[2] + DIMS[2]
    return np.random.default_rng(42).standard_normal(n) * 0.2


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.is_distributed() and parallel.world_size() == world and parallel.rank() == rank
        head = OracleHead(_p0())
        stats_log = []
        calls, real_all_reduce = [], dist.all_reduce

        def counting_all_reduce(t, *a, **k):
            calls.append(t.numel())
            return real_all_reduce(t, *a, **k)
        dist.all_reduce = counting_all_reduce
        for step in range(4):
            x, y = _data(100 + step, 16)
            xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]       # shard the batch across ranks
            stats_log.append(parallel.dp_step(head, xs, ys, lr=1e-2).tolist())
        dist.all_reduce = real_all_reduce
        assert calls == [len(head.p) + 2] * 4, calls          # ONE collective per step: gradients + the two statistics
        t = torch.from_numpy(head.p.copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        if rank == 0:
            out.put((head.p, stats_log, [g.numpy() for g in gathered]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(180)
def test_dp_step_world2_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    p_dp, stats_dp, per_rank = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(per_rank[0], per_rank[1])                   # replicas stay identical
    ref = OracleHead(_p0())
    for step in range(4):
        x, y = _data(100 + step, 16)
        s = parallel.dp_step(ref, x, y, lr=1e-2)                        # single process, merged batch
        assert np.allclose(stats_dp[step], s.tolist(), rtol=1e-12)
    assert np.allclose(p_dp, ref.p, rtol=0, atol=1e-12)


def test_single_process_helpers_are_noops():
    assert not parallel.is_distributed() and parallel.world_size() == 1 and parallel.rank() == 0
    t = torch.ones(3)
    assert parallel.allreduce_sum_(t) is t and parallel.broadcast_(t) is t
