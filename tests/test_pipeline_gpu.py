"""-m gpu: the reference-shaped Python surface end to end on synthetic data: augmentation kernels,
dataset batches, transfer_learn (BASELINE configs[0] counterpart), evaluate_files_*, save/load."""
import os

import numpy as np
import pytest

from tests.util_data import make_fewshot_dataset

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    return make_fewshot_dataset(str(tmp_path_factory.mktemp("fewshot")))


def test_augment_kernel_matches_numpy_semantics(data):
    """mkws_augment_batch vs the reference's augment arithmetic (timeshift / silence / rms-matched mix)."""
    from multilingual_kws_amd.embedding import input_data
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["target"], data["bg_dir"], data["unknown"], unknown_percentage=50.0, seed=3)
    train = ds.init_single_target(input_data.AUTOTUNE, data["train"], is_training=True).shuffle(1000).repeat().batch(256)
    spec, labels = next(iter(train))
    assert spec.shape == (256, 49, 40, 1) and labels.shape == (256,) and spec.is_cuda
    audio = ds.last_audio.cpu().numpy()
    # re-derive every clip on the host from the same draws
    rng = np.random.default_rng(3)
    ds2 = input_data.AudioDataset(ms, ["target"], data["bg_dir"], data["unknown"], unknown_percentage=50.0, seed=3)
    order = []
    while len(order) < 256:
        order.extend(ds2.rng.permutation(5).tolist())
    # same generator stream as _make_batch: sil, unk, mix, shift, unknown idx, bg idx/off, volumes
    nf = 256
    sil = ds2.rng.uniform(0, 1, nf) < 0.10
    unk = ~sil & (ds2.rng.uniform(0, 1, nf) < 0.50)
    mix = ~sil & ~unk & (ds2.rng.uniform(0, 1, nf) < 0.8)
    shift = ds2.rng.integers(-1600, 1600, nf)
    usrc = ds2.rng.integers(0, len(data["unknown"]), nf)
    bidx = ds2.rng.integers(0, 2, nf)
    boff = ds2.rng.integers(0, ds2.background_sizes[bidx] - 16000)
    vol = np.where(sil, ds2.rng.uniform(0, 1, nf), ds2.rng.uniform(0, 0.1, nf)).astype(np.float32)
    tgt = np.stack([input_data._read_wav(f, 16000) for f in data["train"]])
    unkb = np.stack([input_data._read_wav(f, 16000) for f in data["unknown"]])

    def shifted(x, a):
        out = np.zeros(16000, np.float32)
        if a > 0:
            out[a:] = x[:16000 - a]
        else:
            out[:16000 + a] = x[-a:]
        return out

    lab = labels.cpu().numpy()
    for j in range(256):
        bgs = ds2.background_host[bidx[j], boff[j]:boff[j] + 16000]
        if sil[j]:
            exp, el = bgs * vol[j], 0
        elif unk[j]:
            exp, el = shifted(unkb[usrc[j]], shift[j]), 1
        elif mix[j]:
            exp, el = input_data.add_background(shifted(tgt[order[j]], shift[j]), bgs, vol[j]), 2
        else:
            exp, el = shifted(tgt[order[j]], shift[j]), 2
        assert lab[j] == el
        assert np.abs(audio[j] - exp).max() < 2e-6, (j, sil[j], unk[j], mix[j])
    assert abs(sil.mean() - 0.10) < 0.06 and abs(unk.mean() - 0.45) < 0.1       # label mix of Appendix C.3
    # SpecAugment only ever zeroes whole rows/columns, and the untouched entries equal the frontend's output
    clean = input_data.to_micro_spectrogram(ms, ds.last_audio)
    s = spec[..., 0]
    changed = (s != clean)
    assert (s[changed] == 0).all()
    assert 0.4 < float(changed.any(dim=(1, 2)).float().mean()) < 0.85


def test_validation_batches_are_unaugmented(data):
    from multilingual_kws_amd.embedding import input_data
    ms = input_data.standard_microspeech_model_settings(3)
    ds = input_data.AudioDataset(ms, ["target"], data["bg_dir"], data["unknown"], unknown_percentage=50.0, seed=1)
    val = ds.init_single_target(input_data.AUTOTUNE, data["val"], is_training=False).batch(3)
    batches = list(val)
    assert [b[0].shape[0] for b in batches] == [3, 3, 2]
    assert all((b[1] == 2).all() for b in batches)
    specs = torch.cat([b[0] for b in batches])[..., 0].cpu().numpy()
    ref = np.stack([input_data.file2spec(ms, f) for f in data["val"]])
    assert np.array_equal(specs, ref)
    ev = list(ds.eval_with_silence_unknown(input_data.AUTOTUNE, data["val"], label_from_parent_dir=False).batch(64))
    labs = ev[0][1].cpu().numpy()
    assert (labs == 2).sum() == 8 and (labs == 0).sum() == 0 and (labs == 1).sum() == 4      # int(8*.1)=0 silence, int(8*.5)=4 unknown


def test_transfer_learn_contract_and_learning(data, tmp_path):
    """The reference's canonical call (run.py:281-298, tutorial cell 28): 4 epochs x 64 steps x 64 clips."""
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    ms = input_data.standard_microspeech_model_settings(3)
    csv_path = str(tmp_path / "log.csv")
    name, model, details = tl.transfer_learn(
        target="target", train_files=data["train"], val_files=data["val"], unknown_files=data["unknown"],
        num_epochs=4, num_batches=1, batch_size=64, primary_lr=0.001, backprop_into_embedding=False, embedding_lr=0,
        model_settings=ms, base_model_path="synthetic", base_model_output="dense_2", UNKNOWN_PERCENTAGE=50.0,
        bg_datadir=data["bg_dir"], csvlog_dest=csv_path, verbose=0, seed=11)
    assert set(details) == {"num_epochs", "batch_size", "num_batches", "val_accuracy", "target"}
    assert details["num_epochs"] == 4 and details["batch_size"] == 64 and details["num_batches"] == 1 and details["target"] == "target"
    assert name == f"xfer_epochs_4_bs_64_nbs_1_val_acc_{details['val_accuracy']:0.2f}_target_target"
    h = model.history
    assert all(len(h[k]) == 4 for k in ("loss", "accuracy", "val_loss", "val_accuracy"))
    assert h["loss"][-1] < h["loss"][0]                                        # the canonical 4 x 64 steps of 64 clips: loss falls
    assert open(csv_path).read().splitlines()[0] == "epoch,accuracy,loss,val_accuracy,val_loss"
    # predict / evaluate_files_single_target contract
    tpreds, preds = tl.evaluate_files_single_target(data["val"], 2, model, ms)
    assert preds.shape == (8, 3) and np.allclose(preds.sum(1), 1, atol=1e-5) and np.array_equal(tpreds, preds[:, 2])
    upreds, _ = tl.evaluate_files_single_target(data["unknown"], 2, model, ms)
    assert tpreds.mean() > upreds.mean()                                        # target clips score higher on the target class
    mc = tl.evaluate_files_multiclass(data["val"], 2, model, ms)
    assert len(mc["correct"]) + len(mc["incorrect"]) == 8
    # parity of the returned model with the oracles on identical inputs (argmax bit-exact)
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    from multilingual_kws_amd import weights
    specs = np.stack([input_data.file2spec(ms, f) for f in data["val"]])
    ref_emb = EmbeddingOracle(weights.synthetic_blob()).forward(specs).numpy()
    ref_probs, _ = ho.forward(model.head.get_params(), ref_emb)
    assert np.abs(preds - ref_probs).max() < 1e-4 and np.array_equal(preds.argmax(1), ref_probs.argmax(1))
    # save / load round trip
    model.save(str(tmp_path / "m"))
    # (the frozen phase embeds FORWARD_CLIPS // batch_size batches per forward pass: the returned model's handle is planned for 3072 clips)
    assert model.embedding.max_batch == 64 * tl.steps_per_forward(64) == tl.FORWARD_CLIPS == 3072
    again = tl.TransferLearnedModel.load(str(tmp_path / "m"), max_batch=model.embedding.max_batch)       # same handle size = same plan: bit for bit
    assert np.array_equal(again.predict(specs[..., None]), model.predict(specs[..., None]))
    small = tl.TransferLearnedModel.load(str(tmp_path / "m"), max_batch=64)                              # another plan: fp32 round-off
    assert np.abs(small.predict(specs[..., None]) - model.predict(specs[..., None])).max() < 1e-5


def test_transfer_learn_cut_at_another_layer(data, tmp_path):
    """base_model_output names any flat layer of the base model (reference transfer_learning.py:38-42 uses get_layer): the head then
    sits on that layer's features."""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    ms = input_data.standard_microspeech_model_settings(3)
    specs = np.stack([input_data.file2spec(ms, f) for f in data["val"]])
    taps = {}
    EmbeddingOracle(weights.synthetic_blob()).forward(specs, taps)
    for layer, tap, width in (("dense_1", "dense_1", 2048), ("global_average_pooling2d", "gap", 1280)):
        name, model, details = tl.transfer_learn(
            target="target", train_files=data["train"], val_files=data["val"], unknown_files=data["unknown"],
            num_epochs=1, num_batches=1, batch_size=16, primary_lr=0.001, backprop_into_embedding=False, embedding_lr=0,
            model_settings=ms, base_model_path="synthetic", base_model_output=layer, bg_datadir=data["bg_dir"], verbose=0, seed=5)
        assert model.head.in_dim == width and model.head.get_params().shape == (width * 18 + 18 + 18 * 3 + 3,)
        preds = model.predict(specs[..., None])
        ref_probs, _ = ho.forward(model.head.get_params(), taps[tap], in_dim=width)
        assert np.abs(preds - ref_probs).max() < 1e-4 and np.array_equal(preds.argmax(1), ref_probs.argmax(1))
        assert preds.shape == (8, 3) and np.allclose(preds.sum(1), 1, atol=1e-5)
        model.save(str(tmp_path / layer))
        again = tl.TransferLearnedModel.load(str(tmp_path / layer), max_batch=model.embedding.max_batch)   # same handle size = same plan: bit for bit
        assert np.array_equal(again.predict(specs[..., None]), preds)
        small = tl.TransferLearnedModel.load(str(tmp_path / layer), max_batch=16)           # a live-serving handle (cluster plan): fp32 round-off
        assert np.abs(small.predict(specs[..., None]) - preds).max() < 1e-5


def test_smoke_entry_point():
    import __graft_entry__ as g
    g.smoke()


def test_hot_path_is_graph_capturable():
    """include/mkws.h promises: asynchronous on the caller's stream, no allocation and no synchronisation after
    create.  Capture frontend + embedding + 3 heads into one HIP graph and replay it on new input."""
    import torch
    from multilingual_kws_amd import synth, weights
    from multilingual_kws_amd.embedding_model import EmbeddingModel
    from multilingual_kws_amd.frontend import Frontend
    from multilingual_kws_amd.head import Head
    dev = torch.device("cuda:0")
    fe, em = Frontend(max_samples=16000), EmbeddingModel(weights.synthetic_blob(), max_batch=8)
    heads = [Head(max_batch=8, seed=s) for s in (1, 2, 3)]
    a0 = torch.from_numpy(synth.clips_float32(8)).to(dev)
    a1 = torch.from_numpy(synth.clips_float32(8, first_clip=100)).to(dev)

    def run(x):
        return Head.forward_many(heads, em.forward(fe.forward(x)))

    ref0, ref1 = run(a0).clone(), run(a1).clone()
    static_in = a0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(static_in)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run(static_in)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref0)
    static_in.copy_(a1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref1) and not torch.equal(ref0, ref1)
