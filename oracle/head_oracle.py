"""numpy restatement of the few-shot head and its training step
(multilingual_kws/embedding/transfer_learning.py:47-59,86-93): Dense(18,tanh) -> Dense(3,softmax),
SparseCategoricalCrossentropy, Keras Adam (keras/optimizer_v2/adam.py).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED BY THE REFERENCE (no tests/vectors there; Keras is not
installable here): checked against finite differences, hand-computed cases and -- loss, gradients and 50 Adam steps -- a PyTorch
float64 autograd / torch.optim.Adam rendering of the same head (tests/test_oracle_head.py).
Parameter vector layout: W1[in,hid] | b1[hid] | W2[hid,cls] | b2[cls].
"""
import numpy as np


def glorot_uniform_params(in_dim=1024, hidden=18, classes=3, seed=0):
    """Keras Dense default init: kernel glorot_uniform, bias zeros."""
    rng = np.random.default_rng(seed)
    l1 = np.sqrt(6.0 / (in_dim + hidden))
    l2 = np.sqrt(6.0 / (hidden + classes))
    W1 = rng.uniform(-l1, l1, (in_dim, hidden))
    W2 = rng.uniform(-l2, l2, (hidden, classes))
    return np.concatenate([W1.ravel(), np.zeros(hidden), W2.ravel(), np.zeros(classes)]).astype(np.float32)


def unpack(p, in_dim, hidden, classes):
    o = 0
    W1 = p[o:o + in_dim * hidden].reshape(in_dim, hidden); o += in_dim * hidden
    b1 = p[o:o + hidden]; o += hidden
    W2 = p[o:o + hidden * classes].reshape(hidden, classes); o += hidden * classes
    b2 = p[o:o + classes]
    return W1, b1, W2, b2


def forward(p, x, in_dim=1024, hidden=18, classes=3, dtype=np.float64):
    W1, b1, W2, b2 = [a.astype(dtype) for a in unpack(np.asarray(p), in_dim, hidden, classes)]
    x = np.asarray(x, dtype=dtype)
    h = np.tanh(x @ W1 + b1)
    z = h @ W2 + b2
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True), h


def loss_and_grad(p, x, y, in_dim=1024, hidden=18, classes=3, dtype=np.float64):
    """mean sparse CE over the batch, its gradient (flat, same layout as p), #correct."""
    W1, b1, W2, b2 = [a.astype(dtype) for a in unpack(np.asarray(p), in_dim, hidden, classes)]
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y)
    B = x.shape[0]
    probs, h = forward(p, x, in_dim, hidden, classes, dtype)
    loss_rows = -np.log(probs[np.arange(B), y])
    dz = probs.copy()
    dz[np.arange(B), y] -= 1.0
    dz /= B
    dW2 = h.T @ dz
    db2 = dz.sum(0)
    dpre = (dz @ W2.T) * (1.0 - h * h)
    dW1 = x.T @ dpre
    db1 = dpre.sum(0)
    g = np.concatenate([dW1.ravel(), db1, dW2.ravel(), db2])
    return float(loss_rows.mean()), g, int((probs.argmax(1) == y).sum()), float(loss_rows.sum())


class KerasAdam:
    """lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; theta -= lr_t*m/(sqrt(v)+eps)  (eps OUTSIDE the
    bias-corrected sqrt, unlike torch.optim.Adam)."""

    def __init__(self, n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, dtype=np.float64):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = np.zeros(n, dtype)
        self.v = np.zeros(n, dtype)
        self.t = 0

    def step(self, p, g):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        self.m = self.b1 * self.m + (1 - self.b1) * g
        self.v = self.b2 * self.v + (1 - self.b2) * g * g
        return p - lr_t * self.m / (np.sqrt(self.v) + self.eps)
