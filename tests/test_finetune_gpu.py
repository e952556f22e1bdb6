"""-m gpu: BASELINE configs[3] -- one data-parallel fine-tune step at 512 clips per GPU, every stage
(augmentation -> micro-frontend -> SpecAugment -> EfficientNet-B0 embedding -> head loss/gradient -> Keras Adam)
against the CPU oracle chain on the same draws; and the RCCL path (backend "nccl") exercised on the device
Head's own gradient buffer in a world of one."""
import socket

import numpy as np
import pytest

from tests.util_data import make_fewshot_dataset

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    return make_fewshot_dataset(str(tmp_path_factory.mktemp("fewshot512")), n_unknown=64)


def _apply_masks(spec, masks, nf=2, nt=2):
    """spec_augment (reference input_data.py:306-364) on the host: multiplicative zeroing of channel / frame bands."""
    out = spec.copy()
    for b in range(spec.shape[0]):
        m = masks[b]
        for k in range(nf):
            if m[2 * k + 1] > 0:
                out[b, :, m[2 * k]:m[2 * k] + m[2 * k + 1]] = 0
        for k in range(nt):
            if m[2 * (nf + k) + 1] > 0:
                out[b, m[2 * (nf + k)]:m[2 * (nf + k)] + m[2 * (nf + k) + 1], :] = 0
    return out


def test_specaug_with_more_than_two_masks_per_axis(data):
    """mkws_specaug_apply_n with SpecAugParams(frequency_n_range=3, time_n_range=5): the batch equals the host re-derivation from the same draws, bit for bit."""
    import ctypes
    from multilingual_kws_amd import _lib
    from multilingual_kws_amd.embedding import input_data
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["t"], None, [], spec_aug_params=input_data.SpecAugParams(percentage=100, frequency_n_range=3, time_n_range=5,
                                                                                                 frequency_max_px=5, time_max_px=4), seed=2)
    B = 300
    masks = ds._draw_specaug_masks(B)
    spec = (np.random.default_rng(0).integers(1, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256))
    d_spec, d_masks = torch.from_numpy(spec).cuda(), torch.from_numpy(masks).cuda()
    _lib.check(_lib.lib().mkws_specaug_apply_n(ctypes.c_void_p(d_spec.data_ptr()), ctypes.c_void_p(d_masks.data_ptr()), 3, 5, B, 49, 40, _lib.current_stream_ptr()))
    want = _apply_masks(spec, masks, 3, 5)
    assert np.array_equal(d_spec.cpu().numpy(), want) and (want == 0).any() and (masks[:, 5] > 0).any() and (masks[:, 15] > 0).any()


def test_full_finetune_step_512_matches_oracle_chain(data):
    from multilingual_kws_amd import parallel, weights
    from multilingual_kws_amd.embedding import input_data
    from multilingual_kws_amd.embedding_model import EmbeddingModel
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    from oracle.frontend_oracle import FrontendOracle
    B, lr = 512, 1e-3
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["target"], data["bg_dir"], data["unknown"], unknown_percentage=50.0,
                                 spec_aug_params=input_data.SpecAugParams(percentage=80), seed=5)
    it = iter(ds.init_single_target(input_data.AUTOTUNE, data["train"], is_training=True).shuffle(1000).repeat().batch(B))
    blob = weights.synthetic_blob()
    em = EmbeddingModel(blob, max_batch=B)
    p0 = ho.glorot_uniform_params(seed=3)
    head = Head(params=p0, max_batch=B)
    eo, fo = EmbeddingOracle(blob), FrontendOracle()
    p_chain = p0.astype(np.float64)                   # oracle frontend -> oracle embedding -> oracle head / Adam
    p_head = p0.astype(np.float64)                    # oracle head / Adam fed the DEVICE embedding (isolates the head step)
    opt_chain, opt_head = ho.KerasAdam(len(p0), lr=lr), ho.KerasAdam(len(p0), lr=lr)
    for step in range(2):
        spec, labels = next(it)
        assert spec.shape == (B, 49, 40, 1)
        audio, masks, lab = ds.last_audio.cpu().numpy(), ds.last_masks, labels.cpu().numpy()
        assert masks is not None and masks.any()
        # frontend + SpecAugment: bit-exact against the C oracle + host masking (value-level check of mkws_specaug_apply)
        ref_spec = _apply_masks(fo.run_batch_f32(audio), masks)
        assert np.array_equal(spec[..., 0].cpu().numpy(), ref_spec)
        # embedding: north_star tolerance 1e-3 relative (asserted tighter)
        emb = em.forward(spec[..., 0])
        ref_emb = eo.forward(ref_spec).numpy()
        rel = np.abs(emb.cpu().numpy() - ref_emb).max() / np.abs(ref_emb).max()
        assert rel < 1e-4, rel
        # head loss / gradient / Adam through the data-parallel step (world of one: no collective)
        stats = parallel.dp_step(head, emb, labels, lr=lr).tolist()
        _, g_chain, nc, lsum = ho.loss_and_grad(p_chain, ref_emb, lab)
        _, g_head, nc_h, lsum_h = ho.loss_and_grad(p_head, emb.cpu().numpy(), lab)
        assert abs(stats[0] - lsum_h) < 1e-4 * max(1.0, lsum_h) and int(stats[1]) == nc_h
        assert abs(stats[0] - lsum) < 1e-3 * max(1.0, lsum)
        got = head.grad_view(with_stats=True).cpu().numpy()
        assert got.shape == (len(p0) + 2,) and got[-2] == np.float32(stats[0]) and got[-1] == stats[1]   # stats ride behind the gradient
        assert np.abs(got[:-2] - g_head).max() / np.abs(g_head).max() < 2e-5
        assert np.abs(got[:-2] - g_chain).max() / np.abs(g_chain).max() < 1e-3
        p_chain, p_head = opt_chain.step(p_chain, g_chain), opt_head.step(p_head, g_head)
        dev_p = head.get_params()
        assert np.abs(dev_p - p_head).max() < 2e-6
        # whole chain: Adam's first steps are ~lr*sign(g), so entries with |g| ~ eps amplify the 1e-5 embedding
        # difference; bounded well below one step (lr) and negligible in the mean
        d = np.abs(dev_p - p_chain)
        assert d.max() < 0.2 * lr and d.mean() < 1e-3 * lr, (d.max(), d.mean())
    labs = labels.cpu().numpy()
    assert set(np.unique(labs)) <= {0, 1, 2} and (labs == 0).mean() < 0.2 and 0.3 < (labs == 1).mean() < 0.6


def test_dp_step_over_rccl_on_the_device_head():
    """parallel.dp_step with the collective forced in a 1-rank "nccl" (= RCCL) group: the all-reduce runs in place on
    Head.grad_view(with_stats=True) -- the hand-rolled __cuda_array_interface__ alias of the handle's device memory --
    and the trajectory equals the oracle's.  (The N > 1 arithmetic is covered on CPU by tests/test_distributed_cpu.py.)"""
    import torch.distributed as dist
    from multilingual_kws_amd import parallel
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        assert parallel.world_size() == 1 and not parallel.is_distributed()
        rng = np.random.default_rng(0)
        x = torch.from_numpy((rng.standard_normal((512, 1024)) * 0.3).astype(np.float32)).to(dev)
        y = torch.from_numpy(rng.integers(0, 3, 512).astype(np.int32)).to(dev)
        p0 = ho.glorot_uniform_params(seed=7)
        head, plain = Head(params=p0, max_batch=512), Head(params=p0, max_batch=512)
        calls, real = [], dist.all_reduce

        def counting(t, *a, **k):
            calls.append((t.data_ptr(), t.numel()))
            return real(t, *a, **k)
        dist.all_reduce = counting
        opt, p = ho.KerasAdam(len(p0), lr=1e-3), p0.astype(np.float64)
        for step in range(5):
            stats = parallel.dp_step(head, x, y, lr=1e-3, force_collective=True)
            s = stats.tolist()
            parallel.dp_step(plain, x, y, lr=1e-3)
            _, g, nc, lsum = ho.loss_and_grad(p, x.cpu().numpy(), y.cpu().numpy())
            p = opt.step(p, g)
            assert abs(s[0] - lsum) < 1e-4 * lsum and int(s[1]) == nc
        dist.all_reduce = real
        L = head.L
        assert calls == [(L.mkws_head_grads(head.h), len(p0) + 2)] * 5      # ONE collective per step, in place on the handle's buffer
        assert np.abs(head.get_params() - p).max() < 2e-6
        assert np.array_equal(head.get_params(), plain.get_params())        # sum over a world of one is the identity, bit for bit
        # a [B,1024] all_gather of embeddings (optional for inference) also runs on RCCL
        out = [torch.empty_like(x)]
        dist.all_gather(out, x)
        assert torch.equal(out[0], x)
    finally:
        dist.destroy_process_group()


def _fresh(data, seed, B):
    from multilingual_kws_amd.embedding import input_data
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["target"], data["bg_dir"], data["unknown"], unknown_percentage=50.0,
                                 spec_aug_params=input_data.SpecAugParams(percentage=80), seed=seed)
    return ds, ds.init_single_target(input_data.AUTOTUNE, data["train"], is_training=True).shuffle(1000).repeat().batch(B)


def test_grouped_batches_are_the_step_by_step_batches(data):
    """input_data.BatchGroups.take(G): the same clips, labels and SpecAugment masks, in the same order, as G consecutive batches of the
    step-by-step stream, bit for bit (same host draws; augmentation, micro-frontend and SpecAugment are per-clip kernels)."""
    from multilingual_kws_amd.embedding import input_data
    B, G = 96, 4
    (ds1, t1), (ds2, t2) = _fresh(data, 11, B), _fresh(data, 11, B)
    it = iter(t1)
    singles = [next(it) for _ in range(2 * G + 1)]
    audio1 = None
    groups = input_data.BatchGroups(t2)
    got = [groups.take(G), groups.take(G), groups.take(1)]
    spec1, lab1 = torch.cat([s for s, _ in singles]), torch.cat([l for _, l in singles])
    spec2, lab2 = torch.cat([s for s, _ in got]), torch.cat([l for _, l in got])
    assert got[0][0].shape == (G * B, 49, 40, 1) and got[2][0].shape == (B, 49, 40, 1)
    assert torch.equal(spec1, spec2) and torch.equal(lab1, lab2)
    assert (spec1 == 0).any() and len(set(lab1.tolist())) == 3


@pytest.mark.parametrize("overlap", [False, True])
def test_pipelined_head_training_equals_step_by_step(data, overlap):
    """transfer_learning.FrozenHeadTrainer (G optimizer steps per forward pass of the frozen embedding, optimizer steps optionally on a
    second stream) against one forward pass per step: on the SAME handle the embedding rows are bit-identical whatever the batch they
    ride in (plans are per handle), so the head parameters, Adam moments and per-step statistics must be bit-identical too; against a
    handle planned for the small batch (other tile / split plans: fp32 round-off in the embedding) the parameters agree to a small
    fraction of one Adam step.  11 steps with G = 4: groups of 4, 4 and a cut group of 3 (group_limit)."""
    from multilingual_kws_amd import parallel, weights
    from multilingual_kws_amd.embedding import transfer_learning as tl
    from multilingual_kws_amd.embedding_model import EmbeddingModel
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    B, G, steps, lr = 64, 4, 11, 1e-3
    blob = weights.synthetic_blob()
    big, small = EmbeddingModel(blob, max_batch=G * B), EmbeddingModel(blob, max_batch=B)
    p0 = ho.glorot_uniform_params(seed=3)
    # reference legs: a forward pass per optimizer step
    legs = {}
    for name, em in (("same_handle", big), ("small_handle", small)):
        ds, tds = _fresh(data, 21, B)
        head, it, stats = Head(params=p0, max_batch=B), iter(tds), []
        for _ in range(steps):
            spec, labels = next(it)
            stats.append(parallel.dp_step(head, em.forward(spec), labels, lr=lr).tolist())
        legs[name] = (head.state_view().cpu().numpy().copy(), stats)
    # pipelined
    ds, tds = _fresh(data, 21, B)
    head = Head(params=p0, max_batch=B)
    ft = tl.FrozenHeadTrainer(big, head, tds, B, lr, group=G, overlap=overlap)
    acc, stats = torch.zeros(2, dtype=torch.float64, device="cuda"), []
    for i in range(steps):
        st = ft.step(group_limit=steps - i)
        ft.accumulate(acc, st)
        if overlap:
            ft.side.synchronize()
        stats.append(st.tolist())
    ft.finish()
    torch.cuda.synchronize()
    assert ft.G == G and ft.forwards == 3
    state = head.state_view().cpu().numpy()
    assert np.array_equal(state, legs["same_handle"][0])                       # params | grads | m | v, bit for bit
    assert stats == legs["same_handle"][1]
    assert np.allclose(acc.cpu().numpy(), np.sum(np.asarray(stats, dtype=np.float64), axis=0), rtol=0, atol=1e-9)
    n = len(p0)
    d = np.abs(state[:n] - legs["small_handle"][0][:n])
    assert d.max() < 0.2 * lr and d.mean() < 2e-3 * lr, (d.max(), d.mean())
    assert [s[1] for s in stats] == [s[1] for s in legs["small_handle"][1]]    # the same rows classified correctly at every step


def test_transfer_learn_groups_steps_of_the_frozen_phase(data):
    """transfer_learn's frozen phase runs FORWARD_CLIPS // batch_size optimizer steps per forward pass and never carries a group across
    an epoch boundary (validation reads the head there): 2 epochs x (4 x 4 =) 16 steps of 4 clips at FORWARD_CLIPS = 24 -> groups of 6, 6, 4
    per epoch; the returned history is that of the same seeds with a forward pass per step."""
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    ms = input_data.standard_microspeech_model_settings(3)
    kw = dict(target="target", train_files=data["train"], val_files=data["val"], unknown_files=data["unknown"], num_epochs=2, num_batches=4,
              batch_size=4, primary_lr=1e-3, backprop_into_embedding=False, embedding_lr=0, model_settings=ms, base_model_path="synthetic",
              base_model_output="dense_2", bg_datadir=data["bg_dir"], verbose=0, seed=4)
    seen, real = [], tl.FrozenHeadTrainer._refill

    def spy(self, g):
        seen.append(g)
        return real(self, g)
    old = tl.FORWARD_CLIPS
    tl.FrozenHeadTrainer._refill = spy
    try:
        tl.FORWARD_CLIPS = 24
        _, m_grouped, _ = tl.transfer_learn(**kw)
        assert seen == [6, 6, 4, 6, 6, 4]
        del seen[:]
        tl.FORWARD_CLIPS = 4
        _, m_single, _ = tl.transfer_learn(**kw)
        assert seen == [1] * 32
    finally:
        tl.FrozenHeadTrainer._refill, tl.FORWARD_CLIPS = real, old
    # both calls run a 64-clip handle (max_batch = max(group * batch, 64)): same plan, same bits
    assert np.array_equal(m_grouped.head.get_params(), m_single.head.get_params())
    assert m_grouped.history == m_single.history
