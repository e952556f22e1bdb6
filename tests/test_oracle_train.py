"""Training-mode oracle (groundwork for SURVEY.md section 8f-4): eval-mode equality with the inference oracle, batch-norm
batch statistics against a hand computation, autograd against central finite differences (float64)."""
import numpy as np
import torch

from multilingual_kws_amd import weights
from oracle.efficientnet_oracle import EmbeddingOracle
from oracle.efficientnet_train_oracle import BN_MOMENTUM, TrainableEmbeddingOracle


def _spec(rng, n):
    return (rng.integers(0, 670, size=(n, 49, 40)).astype(np.float32) * np.float32(10 / 256))


def test_eval_mode_equals_the_inference_oracle():
    blob = weights.synthetic_blob()
    spec = _spec(np.random.default_rng(0), 3)
    a = EmbeddingOracle(blob, torch.float64).forward(spec).numpy()
    b = TrainableEmbeddingOracle(blob).forward(spec, training=False).detach().numpy()
    assert np.allclose(a, b, rtol=1e-12, atol=1e-12)


def test_training_mode_uses_batch_statistics_and_updates_moving_averages():
    blob = weights.synthetic_blob()
    o = TrainableEmbeddingOracle(blob)
    spec = _spec(np.random.default_rng(1), 4)
    e_train = o.forward(spec, training=True).detach().numpy()
    e_eval = o.forward(spec, training=False).detach().numpy()
    assert not np.allclose(e_train, e_eval)                                 # different normalisation statistics
    # stem BN: batch statistics of the stem conv output, momentum 0.99
    x = torch.as_tensor(spec, dtype=torch.float64)[:, None] / 255.0
    x = torch.nn.functional.pad(x, (0, 1, 1, 1))
    y = torch.nn.functional.conv2d(x, o.w["stem_conv/kernel"].detach().permute(3, 2, 0, 1), stride=2)
    mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False)
    assert torch.allclose(o.new_moving["stem_bn/moving_mean"], BN_MOMENTUM * o.w["stem_bn/moving_mean"] + (1 - BN_MOMENTUM) * mean)
    n = y.shape[0] * y.shape[2] * y.shape[3]          # Keras' fused BN feeds the Bessel-corrected variance to the moving average
    assert torch.allclose(o.new_moving["stem_bn/moving_variance"], BN_MOMENTUM * o.w["stem_bn/moving_variance"] + (1 - BN_MOMENTUM) * var * n / (n - 1))
    assert len(o.new_moving) == 2 * 49                                      # every BatchNormalization layer (49 of them)


def test_autograd_matches_finite_differences():
    blob = weights.synthetic_blob()
    o = TrainableEmbeddingOracle(blob)
    spec = _spec(np.random.default_rng(2), 2)
    proj = torch.from_numpy(np.random.default_rng(3).standard_normal(1024))

    def loss():
        return (o.forward(spec, training=True) @ proj).sum()

    o.zero_grad()
    loss().backward()
    g = o.grads()
    assert all(v is not None for v in g.values()) and len(g) == len(o.trainable)
    rng = np.random.default_rng(4)
    for name in ("stem_conv/kernel", "block2a_dwconv/depthwise_kernel", "block4b_se_reduce/kernel", "block6c_project_bn/gamma",
                 "top_conv/kernel", "dense_1/bias"):
        w = o.w[name]
        idx = tuple(int(rng.integers(0, d)) for d in w.shape)
        with torch.no_grad():
            old = w[idx].item()
            h = 1e-5 * max(1.0, abs(old))
            w[idx] = old + h; lp = loss().item()
            w[idx] = old - h; lm = loss().item()
            w[idx] = old
        fd = (lp - lm) / (2 * h)
        assert abs(fd - g[name][idx]) <= 1e-5 * max(1.0, abs(fd)), (name, fd, g[name][idx])
