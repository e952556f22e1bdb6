"""-m gpu: the embedding-training loop (reference multilingual_kws/train_multilingual_embedding.py) -- the logits layer and its
from-logits cross-entropy against float64 torch, then the loop's contract: checkpoint naming / save_best_only, CSV log, history,
resume, and the saved checkpoint as `transfer_learn(base_model_path=...)` input."""
import csv
import glob
import os
import pickle

import numpy as np
import pytest

from tests.util_data import tone_clip, write_wav

torch = pytest.importorskip("torch")


def test_checkpoint_name_is_keras_formatting():
    from multilingual_kws_amd.train_multilingual_embedding import checkpoint_name
    assert checkpoint_name("multilingual_context_", 20, 0.70584) == "multilingual_context_.020-0.7058"
    assert checkpoint_name("m", 3, 1.0) == "m.003-1.0000"


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(64, 761), (5, 4), (37, 300)])
def test_logits_layer_and_softmax_ce_against_float64(B, N):
    from multilingual_kws_amd.train_multilingual_embedding import LogitsLayer
    rng = np.random.default_rng(B + N)
    emb = (rng.standard_normal((B, 1024)) * 0.5).astype(np.float32)
    y = rng.integers(0, N, B)
    lay = LogitsLayer(N, seed=1)
    p0 = lay.params.cpu().numpy()
    W = torch.tensor(p0[:1024 * N].reshape(1024, N), dtype=torch.float64, requires_grad=True)
    b = torch.tensor(p0[1024 * N:], dtype=torch.float64, requires_grad=True)
    x = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    z = x @ W + b
    loss = torch.nn.functional.cross_entropy(z, torch.from_numpy(y))
    loss.backward()
    dev = lay.device
    logits = lay.forward(torch.from_numpy(emb).to(dev)).cpu().numpy()
    assert np.abs(logits - z.detach().numpy()).max() < 1e-4
    stats, d_emb = lay.loss_grad(torch.from_numpy(emb).to(dev), torch.from_numpy(y.astype(np.int32)).to(dev))
    st = stats.tolist()
    assert abs(st[0] / B - float(loss)) < 1e-5 * max(1.0, float(loss)) and st[1] == float((z.argmax(1).numpy() == y).sum())
    rel = lambda a, r: float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))
    assert rel(lay.dW.cpu().numpy().reshape(1024, N), W.grad.numpy()) < 1e-4
    assert rel(lay.db.cpu().numpy(), b.grad.numpy()) < 1e-4
    assert rel(d_emb.cpu().numpy(), x.grad.numpy()) < 1e-4
    # deterministic: the same call again gives the same bits
    g1 = lay.grads.clone()
    lay.loss_grad(torch.from_numpy(emb).to(dev), torch.from_numpy(y.astype(np.int32)).to(dev))
    assert torch.equal(lay.grads, g1)


@pytest.mark.gpu
def test_embedding_training_loop_contract(tmp_path):
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from multilingual_kws_amd.train_multilingual_embedding import EmbeddingClassifier, train_embedding
    rng = np.random.default_rng(0)
    words = {"uno": 500, "dos": 1300, "tres": 2600}
    train, val = [], []
    for w, f in words.items():
        for i in range(10):
            p = str(tmp_path / "data" / w / f"{w}{i}.wav")
            write_wav(p, tone_clip(f + 15 * rng.standard_normal(), rng, burst=(2000, 12000)))
            (train if i < 8 else val).append(p)
    bg = tmp_path / "_background_noise_"
    write_wav(str(bg / "n.wav"), (600 * rng.standard_normal(16000 * 4)).astype(np.int16))
    out = tmp_path / "models"
    with pytest.raises(ValueError, match="create model dir"):
        train_embedding(list(words), train, val, str(bg), str(out), epochs=1)
    os.makedirs(out)
    with pytest.raises(ValueError):
        train_embedding(list(words), train, val, str(tmp_path / "nope"), str(out), epochs=1)
    model, hist = train_embedding(list(words), train, val, str(bg), str(out), epochs=4, batch_size=8, learning_rate=1e-3, basename="ctx_", verbose=0, seed=3)
    assert set(hist) == {"loss", "accuracy", "val_loss", "val_accuracy"} and all(len(v) == 4 and np.isfinite(v).all() for v in hist.values())
    assert model.num_labels == 4 and model.commands == ["_silence_", "uno", "dos", "tres"]
    assert min(hist["loss"][1:]) < hist["loss"][0]
    # ModelCheckpoint(save_best_only, monitor val_accuracy, mode max): one directory per improvement, named epoch (1-based) and val_accuracy
    ck = sorted(os.path.basename(p) for p in glob.glob(str(out / "ctx_.*")))
    improvements, best = [], -1.0
    for e, va in enumerate(hist["val_accuracy"]):
        if va > best:
            best = va
            improvements.append(f"ctx_.{e + 1:03d}-{va:.4f}")
    assert ck == sorted(improvements) and len(ck) >= 1
    rows = list(csv.reader(open(out / "ctx__log_0.csv")))
    assert rows[0] == ["epoch", "accuracy", "loss", "val_accuracy", "val_loss"] and [r[0] for r in rows[1:]] == ["0", "1", "2", "3"]
    assert pickle.load(open(out / "history_keras_0.pkl", "rb")) == hist
    # the last checkpoint: logits of the validation clips reproduce through load(); the embedding part feeds transfer_learn
    last = str(out / sorted(improvements)[-1])
    again = EmbeddingClassifier.load(last)
    ms = input_data.standard_microspeech_model_settings(4)
    specs = np.stack([input_data.file2spec(ms, f) for f in val])
    z = again.predict(specs[..., None])
    assert z.shape == (6, 4) and np.isfinite(z).all()
    emb_model, blob = tl.load_base_model(last, max_batch=8)
    assert blob.shape == (weights.weight_count(),) and not np.array_equal(blob, weights.synthetic_blob(3, calibrate=False))
    from oracle.efficientnet_oracle import EmbeddingOracle
    ref = EmbeddingOracle(blob).forward(specs).numpy()
    got = emb_model.forward(torch.from_numpy(specs).cuda()).cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-3
    # resume (the reference re-loads `multilingual_context_.020-0.7058` and keeps training): a second log / history index, training continues
    model2, hist2 = train_embedding(list(words), train, val, str(bg), str(out), epochs=1, batch_size=8, base_checkpoint=last, basename="ctx_", verbose=0, seed=4,
                                    steps_per_epoch=2)
    assert os.path.isfile(out / "ctx__log_1.csv") and os.path.isfile(out / "history_keras_1.pkl") and len(hist2["loss"]) == 1
    with pytest.raises(ValueError, match="labels"):
        train_embedding(["uno", "dos"], train[:8], val[:2], str(bg), str(out), epochs=1, base_checkpoint=last, verbose=0)
