#!/usr/bin/env python
"""Generates tests/golden/keras_golden.npz -- the vectors that pin this repo's oracles (and kernels) to the REAL
TensorFlow / Keras code path the reference executes.

RUN THIS WHERE TENSORFLOW (>= 2.4, the reference pins 2.7: docker/Dockerfile:1) IS INSTALLED.  TensorFlow is not
installable in the MI355X image, so this script cannot run there and the file it writes is absent until someone
runs it; tests/test_keras_golden.py skips loudly while it is absent and consumes it as soon as it is committed:

    python tools/make_keras_golden.py            # writes tests/golden/keras_golden.npz (about 1 MB)

It needs only tensorflow + numpy (not the built HIP library: the tensor order comes from
tools/embedding_manifest.json).  What it records:

  * micro-frontend: raw uint16 [49,40] of `frontend_op.audio_microfrontend` called exactly as
    multilingual_kws/embedding/input_data.py:25-33 does (sample_rate / window_size / window_step / num_channels
    given, every other argument at the TF wrapper's default -- so PCAN at ITS default, SURVEY risk R1) with
    out_type=uint16, for the SURVEY Appendix D.3 signals and the three speech clips under tests/golden/;
    plus the same through the float path (audio float32 -> tf.cast(audio*32768, int16), risk R4);
  * embedding: the Keras model of multilingual_kws/train_multilingual_embedding.py:58-83
    (EfficientNetB0(include_top=False, weights=None, input_shape=(49,40,1)) -> GAP -> Dense 2048 relu ->
    Dense 2048 relu -> Dense 1024 selu) with this repo's synthetic seed-1234 weights assigned INTO the Keras
    variables by name; outputs of the stem, of one block per stage and of dense_2 for 4 fixed spectrograms;
  * head: one Keras-Adam step of Dense(18,tanh) -> Dense(3,softmax) with SparseCategoricalCrossentropy on a fixed
    batch (transfer_learning.py:47-59): loss, gradients and updated parameters.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "keras_golden.npz")
TAP_LAYERS = ["stem_activation", "block1a_project_bn", "block2b_add", "block3b_add", "block4c_add", "block5c_add", "block6d_add",
              "block7a_project_bn", "top_activation"]


def d3_signals():
    """SURVEY Appendix D.3 inputs (int16, 16000 samples)."""
    t = np.arange(16000)
    sig = {"square4": np.tile(np.array([0, 32767, 0, -32768], dtype=np.int64), 4000),
           "sine1k": np.round(16384 * np.sin(2 * np.pi * 1000 * t / 16000.0)).astype(np.int64),
           "zeros": np.zeros(16000, dtype=np.int64)}
    x, lcg = 1, np.zeros(16000, dtype=np.int64)
    for i in range(16000):
        x = (1103515245 * x + 12345) & 0x7FFFFFFF
        lcg[i] = ((((x >> 8) & 0xFFFF) - 32768) * 8192) // 32768
    sig["lcg"] = lcg
    return {k: v.astype(np.int16) for k, v in sig.items()}


def read_wav_pcm16(path, n=16000):
    raw = open(path, "rb").read()
    i = raw.index(b"data")
    size = int.from_bytes(raw[i + 4:i + 8], "little")
    pcm = np.frombuffer(raw[i + 8:i + 8 + size], dtype="<i2")[:n]
    out = np.zeros(n, dtype=np.int16)
    out[:pcm.shape[0]] = pcm
    return out


def main():
    import tensorflow as tf
    from tensorflow.keras import layers, models
    from tensorflow.lite.experimental.microfrontend.python.ops import audio_microfrontend_op as frontend_op
    from multilingual_kws_amd import weights
    out = {"tf_version": np.asarray(tf.__version__)}

    # ---- micro-frontend ----
    sigs = d3_signals()
    for k in range(3):
        sigs[f"tutorial_clip{k}"] = read_wav_pcm16(os.path.join(ROOT, "tests", "golden", f"tutorial_clip{k}.wav"))
    for name, pcm in sigs.items():
        raw = frontend_op.audio_microfrontend(tf.constant(pcm, tf.int16), sample_rate=16000, window_size=30.0, window_step=20.0,
                                              num_channels=40, out_scale=1, out_type=tf.uint16).numpy()
        out[f"fe_in/{name}"], out[f"fe_raw/{name}"] = pcm, raw.astype(np.uint16)
        audio = pcm.astype(np.float32) / np.float32(32768.0)
        i16 = tf.cast(tf.multiply(tf.constant(audio), 32768), tf.int16)          # input_data.py:23
        f32 = frontend_op.audio_microfrontend(i16, sample_rate=16000, window_size=30.0, window_step=20.0, num_channels=40,
                                              out_scale=1, out_type=tf.float32)
        out[f"fe_spec/{name}"] = tf.multiply(f32, 10.0 / 256.0).numpy()           # input_data.py:34

    # ---- embedding model with the synthetic weights ----
    tensors = json.load(open(os.path.join(ROOT, "tools", "embedding_manifest.json")))["tensors"]
    blob = weights.synthetic_blob(1234, tensors=tensors)
    base = tf.keras.applications.EfficientNetB0(include_top=False, weights=None, input_tensor=None, input_shape=(49, 40, 1), pooling=None)
    x = layers.GlobalAveragePooling2D()(base.output)
    x = layers.Dense(2048, activation="relu")(x)
    x = layers.Dense(2048, activation="relu")(x)
    x = layers.Dense(1024, kernel_initializer="lecun_normal", activation="selu")(x)
    model = models.Model(inputs=base.input, outputs=x)
    by_name = {}
    for v in model.variables:
        parts = v.name.split(":")[0].split("/")
        by_name["/".join(parts[-2:])] = v
    assigned = 0
    for t in tensors:
        val = blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"])
        v = by_name.get(t["name"])
        if v is None and t["name"].startswith("normalization/"):
            continue                   # Keras <= 2.7 Normalization(mean, variance, count); un-adapted = identity
        if v is None:
            raise SystemExit(f"Keras model has no variable {t['name']}")
        if tuple(v.shape) != tuple(val.shape):
            raise SystemExit(f"{t['name']}: Keras shape {tuple(v.shape)} != {tuple(val.shape)}")
        v.assign(val)
        assigned += 1
    rng = np.random.default_rng(20260927)
    spec = (rng.integers(0, 670, size=(4, 49, 40, 1)).astype(np.float32) * np.float32(10 / 256))
    names = [n for n in TAP_LAYERS if any(l.name == n for l in model.layers)]
    tap_model = models.Model(inputs=model.inputs, outputs=[model.get_layer(n).output for n in names] + [model.output])
    vals = tap_model.predict(spec)
    out["emb_spec"], out["emb_weights_seed"], out["emb_assigned"] = spec, np.asarray(1234), np.asarray(assigned)
    for n, v in zip(names + ["dense_2"], vals):
        out[f"emb_tap/{n}"] = np.asarray(v, dtype=np.float32)

    # ---- head: one Adam step ----
    hrng = np.random.default_rng(7)
    emb = (hrng.standard_normal((32, 1024)) * 0.3).astype(np.float32)
    y = hrng.integers(0, 3, 32).astype(np.int64)
    head = models.Sequential([layers.Dense(18, activation="tanh", input_shape=(1024,)), layers.Dense(3, activation="softmax")])
    head.compile(optimizer=tf.keras.optimizers.Adam(learning_rate=0.001), loss=tf.keras.losses.SparseCategoricalCrossentropy(), metrics=["accuracy"])
    p0 = np.concatenate([w.numpy().ravel() for w in head.trainable_variables])       # W1 | b1 | W2 | b2
    with tf.GradientTape() as tape:
        loss = head.compiled_loss(tf.constant(y), head(tf.constant(emb), training=True))
    grads = tape.gradient(loss, head.trainable_variables)
    g0 = np.concatenate([g.numpy().ravel() for g in grads])
    head.optimizer.apply_gradients(zip(grads, head.trainable_variables))
    p1 = np.concatenate([w.numpy().ravel() for w in head.trainable_variables])
    out.update({"head_emb": emb, "head_labels": y, "head_p0": p0, "head_loss": np.asarray(float(loss)), "head_grad": g0, "head_p1": p1,
                "head_probs": head.predict(emb)})
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays (TensorFlow {tf.__version__}); commit it -- tests/test_keras_golden.py picks it up")


if __name__ == "__main__":
    main()
