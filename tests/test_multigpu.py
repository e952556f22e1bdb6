"""-m gpu tests that need TWO devices: they skip on the 1-GPU boxes this project is developed on and arm themselves the first time the
suite meets a node (SURVEY row e: one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI).
  (a) parallel.dp_step on device Heads, 2 ranks: ONE all-reduce per step, replicas identical, equal to the single-process step on the
      merged batch (the assertion of tests/test_distributed_cpu.py, on RCCL and the HIP head);
  (b) EmbeddingTrainer.backward(allreduce=True), 2 ranks: the range-wise overlapped gradient all-reduce equals the gradient of the
      merged batch computed by one rank, up to fp32 round-off (BatchNorm statistics are per replica, so (b) compares the all-reduced
      sum with the sum of the two shard gradients computed one after the other on a single device)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs 2 GPUs (arms itself on a multi-GPU node)")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    return dist


def _head_data(step, n):
    rng = np.random.default_rng(500 + step)
    return (rng.standard_normal((n, 1024)) * 0.3).astype(np.float32), rng.integers(0, 3, n).astype(np.int32)


def _dp_worker(rank, world, port, out):
    dist = _init(rank, world, port)
    try:
        from multilingual_kws_amd import parallel
        from multilingual_kws_amd.head import Head
        from oracle import head_oracle as ho
        head = Head(params=ho.glorot_uniform_params(seed=4), max_batch=64)
        calls, real = [], dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        dist.all_reduce = counting
        stats = []
        for step in range(4):
            x, y = _head_data(step, 64)
            sl = slice(rank * 32, (rank + 1) * 32)
            s = parallel.dp_step(head, torch.from_numpy(x[sl]).cuda(), torch.from_numpy(y[sl]).cuda(), lr=1e-2)
            stats.append(s.cpu().tolist())
        dist.all_reduce = real
        assert calls == [head.nparams + 2] * 4, calls                  # ONE collective per step
        params = head.get_params()
        gathered = [None] * world
        dist.all_gather_object(gathered, params)
        if rank == 0:
            out.put((params, stats, gathered))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_step_two_ranks_over_rccl_equals_merged_batch():
    import torch.multiprocessing as mp
    from multilingual_kws_amd import parallel
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    params, stats, per_rank = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(per_rank[0], per_rank[1])                      # replicas stay bit-identical
    ref = Head(params=ho.glorot_uniform_params(seed=4), max_batch=64)
    for step in range(4):
        x, y = _head_data(step, 64)
        s = parallel.dp_step(ref, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), lr=1e-2)
        assert np.allclose(stats[step], s.cpu().tolist(), rtol=1e-5)
    assert np.abs(params - ref.get_params()).max() < 2e-6                           # summation order differs (two 32-row partial sums vs one of 64)


def _trainer_inputs():
    rng = np.random.default_rng(9)
    spec = (rng.integers(0, 670, size=(8, 49, 40)).astype(np.float32) * np.float32(10 / 256))
    return spec, rng.standard_normal((8, 1024)).astype(np.float32)


def _trainer_worker(rank, world, port, out):
    dist = _init(rank, world, port)
    try:
        from multilingual_kws_amd import weights
        from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
        spec, d_emb = _trainer_inputs()
        sl = slice(rank * 4, (rank + 1) * 4)
        tr = EmbeddingTrainer(weights.synthetic_blob())
        tr.forward_train(torch.from_numpy(spec[sl]).cuda(), None)
        tr.backward(torch.from_numpy(d_emb[sl]).cuda(), allreduce=True)
        torch.cuda.synchronize()
        g = {k: v.copy() for k, v in tr.named_grads().items()}
        if rank == 0:
            out.put(g)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_embedding_gradient_allreduce_two_ranks_equals_the_sum_of_the_shards():
    import torch.multiprocessing as mp
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spec, d_emb = _trainer_inputs()
    want = None
    for r in range(2):                                                   # the two shards, one after the other, on one device
        tr = EmbeddingTrainer(weights.synthetic_blob())
        sl = slice(r * 4, (r + 1) * 4)
        tr.forward_train(torch.from_numpy(spec[sl]).cuda(), None)
        tr.backward(torch.from_numpy(d_emb[sl]).cuda())
        g = tr.named_grads()
        want = {k: v.astype(np.float64) for k, v in g.items()} if want is None else {k: want[k] + g[k] for k in want}
    gmax = max(float(np.abs(v).max()) for v in want.values())
    for k, v in want.items():
        assert np.abs(got[k] - v).max() <= 1e-5 * max(float(np.abs(v).max()), 1e-3 * gmax), k


def _grouped_worker(rank, world, port, out, root):
    dist = _init(rank, world, port)
    try:
        from multilingual_kws_amd import weights
        from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
        from multilingual_kws_amd.embedding_model import EmbeddingModel
        from multilingual_kws_amd.head import Head
        from oracle import head_oracle as ho
        from tests.util_data import make_fewshot_dataset
        data = make_fewshot_dataset(os.path.join(root, f"rank{rank}"), n_unknown=16)
        ms = input_data.standard_microspeech_model_settings(3)
        em = EmbeddingModel(weights.synthetic_blob(), max_batch=64)
        res = {}
        for overlap in (False, True):
            ds = input_data.AudioDataset(ms, ["target"], data["bg_dir"], data["unknown"], unknown_percentage=50.0,
                                         spec_aug_params=input_data.SpecAugParams(percentage=80), seed=7 + 1000003 * rank)
            tds = ds.init_single_target(input_data.AUTOTUNE, data["train"], is_training=True).shuffle(1000).repeat().batch(16)
            head = Head(params=ho.glorot_uniform_params(seed=4), max_batch=16)
            calls, real = [], dist.all_reduce

            def counting(t, *a, **k):
                calls.append(t.numel())
                return real(t, *a, **k)
            dist.all_reduce = counting
            ft = tl.FrozenHeadTrainer(em, head, tds, 16, 1e-2, group=4, overlap=overlap)
            for i in range(10):
                ft.step(group_limit=10 - i)
            ft.finish()
            torch.cuda.synchronize()
            dist.all_reduce = real
            assert calls == [head.nparams + 2] * 10 and ft.forwards == 3, (calls, ft.forwards)      # one collective per OPTIMIZER step: groups 4, 4, 2
            res[overlap] = head.get_params()
        assert np.array_equal(res[False], res[True])                                                 # the second stream changes nothing
        gathered = [None] * world
        dist.all_gather_object(gathered, res[True])
        if rank == 0:
            out.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_grouped_fine_tune_two_ranks_keeps_one_collective_per_step_and_identical_replicas(tmp_path):
    """transfer_learning.FrozenHeadTrainer under 2 ranks over RCCL: G optimizer steps share a forward pass, every step still does its OWN
    all-reduce of [gradients | loss sum | #correct] (on the second stream when overlap is on), the replicas end bit-identical, and the
    second stream changes nothing."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_grouped_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    per_rank = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(per_rank[0], per_rank[1])
