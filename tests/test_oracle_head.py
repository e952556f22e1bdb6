"""Checks on oracle/head_oracle.py: analytic gradient vs central differences, Keras-Adam arithmetic."""
import numpy as np

from oracle import head_oracle as ho


def test_gradient_matches_finite_differences():
    rng = np.random.default_rng(0)
    dims = (12, 5, 3)
    n = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + dims[2]
    p = rng.standard_normal(n) * 0.3
    x = rng.standard_normal((7, dims[0]))
    y = rng.integers(0, 3, 7)
    loss, g, _, lsum = ho.loss_and_grad(p, x, y, *dims)
    assert abs(lsum - loss * 7) < 1e-12
    for i in rng.choice(n, 25, replace=False):
        d = np.zeros(n); d[i] = 1e-6
        num = (ho.loss_and_grad(p + d, x, y, *dims)[0] - ho.loss_and_grad(p - d, x, y, *dims)[0]) / 2e-6
        assert abs(num - g[i]) < 1e-7 * max(1.0, abs(g[i])), i


def test_keras_adam_first_steps_by_hand():
    opt = ho.KerasAdam(2, lr=0.1, beta1=0.9, beta2=0.999, eps=1e-7)
    p = np.array([1.0, -2.0])
    g = np.array([0.5, -0.25])
    p1 = opt.step(p, g)
    # t=1: m = 0.1 g, v = 0.001 g^2, lr_t = lr*sqrt(0.001)/0.1 -> step = lr * g/|g| (up to eps)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = p - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-7)
    assert np.allclose(p1, exp, rtol=0, atol=1e-15)
    assert np.allclose(p - p1, 0.1 * np.sign(g), atol=1e-5)
    p2 = opt.step(p1, g)
    m2, v2 = 0.9 * 0.1 * g + 0.1 * g, 0.999 * 0.001 * g * g + 0.001 * g * g
    lr_2 = 0.1 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert np.allclose(p2, p1 - lr_2 * m2 / (np.sqrt(v2) + 1e-7), atol=1e-15)


def test_init_and_probabilities():
    p = ho.glorot_uniform_params(1024, 18, 3, seed=1)
    W1, b1, W2, b2 = ho.unpack(p, 1024, 18, 3)
    assert p.shape == (18507,) and np.all(b1 == 0) and np.all(b2 == 0)
    assert np.abs(W1).max() <= np.sqrt(6 / (1024 + 18)) and np.abs(W2).max() <= np.sqrt(6 / 21)
    probs, _ = ho.forward(p, np.random.default_rng(0).standard_normal((5, 1024)))
    assert np.allclose(probs.sum(1), 1) and (probs > 0).all()


def test_head_oracle_against_pytorch_autograd_and_adam():
    """A pin of the head oracle that this project did not write: the same head in PyTorch (torch.nn.functional linear / tanh / cross_entropy in
    float64, gradients by autograd) and torch.optim.Adam.  Keras' Adam (optimizer_v2/adam.py: theta -= lr_t m / (sqrt(v) + eps) with lr_t =
    lr sqrt(1 - b2^t) / (1 - b1^t)) and PyTorch's (theta -= lr / (1 - b1^t) m / (sqrt(v) / sqrt(1 - b2^t) + eps)) are the same recurrence
    up to where eps sits: Keras(eps) == PyTorch(eps / sqrt(1 - b2^t)) at step t, so the torch optimizer's eps is set per step.  50 steps on fresh
    batches: loss, every gradient and every parameter agree to float64 round-off -- m / v recurrences, bias correction and update included."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    dims = (40, 18, 3)
    n = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + dims[2]
    p = ho.glorot_uniform_params(*dims, seed=5).astype(np.float64)
    p[dims[0] * dims[1]:dims[0] * dims[1] + dims[1]] = 0.1 * rng.standard_normal(dims[1])         # non-zero biases
    W1, b1, W2, b2 = [torch.tensor(a.copy(), dtype=torch.float64, requires_grad=True) for a in ho.unpack(p, *dims)]
    lr, beta1, beta2, eps = 1e-2, 0.9, 0.999, 1e-7
    topt = torch.optim.Adam([W1, b1, W2, b2], lr=lr, betas=(beta1, beta2), eps=eps)
    opt = ho.KerasAdam(n, lr=lr, beta1=beta1, beta2=beta2, eps=eps)
    for t in range(1, 51):
        x = rng.standard_normal((16, dims[0])) * 0.5
        y = rng.integers(0, 3, 16)
        loss, g, ncorrect, lsum = ho.loss_and_grad(p, x, y, *dims)
        topt.zero_grad()
        xt = torch.tensor(x, dtype=torch.float64)
        logits = F.linear(torch.tanh(F.linear(xt, W1.T, b1)), W2.T, b2)
        tloss = F.cross_entropy(logits, torch.tensor(y, dtype=torch.long))            # mean sparse CE from logits = Keras' softmax + SparseCategoricalCrossentropy
        tloss.backward()
        tg = np.concatenate([W1.grad.numpy().ravel(), b1.grad.numpy(), W2.grad.numpy().ravel(), b2.grad.numpy()])
        assert abs(float(tloss) - loss) < 1e-12 and np.abs(tg - g).max() < 1e-13, t
        assert ncorrect == int((logits.argmax(1).numpy() == y).sum())
        probs, _ = ho.forward(p, x, *dims)
        assert np.abs(probs - torch.softmax(logits, 1).detach().numpy()).max() < 1e-14
        for grp in topt.param_groups:
            grp["eps"] = eps / np.sqrt(1.0 - beta2 ** t)
        topt.step()
        p = opt.step(p, g)
        tp = np.concatenate([W1.detach().numpy().ravel(), b1.detach().numpy(), W2.detach().numpy().ravel(), b2.detach().numpy()])
        assert np.abs(tp - p).max() < 1e-12, t
    assert np.abs(p - ho.glorot_uniform_params(*dims, seed=5)).max() > 0.05                      # the 50 steps moved the parameters
