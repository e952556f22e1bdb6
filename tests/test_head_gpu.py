"""-m gpu: the few-shot head (forward, loss/gradient, Keras-Adam) against the numpy oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _mk(dev, n, seed=0, scale=0.3):
    rng = np.random.default_rng(seed)
    return (torch.from_numpy((rng.standard_normal((n, 1024)) * scale).astype(np.float32)).to(dev),
            torch.from_numpy(rng.integers(0, 3, n).astype(np.int32)).to(dev))


@pytest.mark.parametrize("B", [1, 3, 64, 300, 512])
def test_forward_loss_and_gradient(dev, B):
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    p0 = ho.glorot_uniform_params(seed=B)
    hd = Head(params=p0, max_batch=512)
    x, y = _mk(dev, B, seed=B)
    probs = hd.forward(x).cpu().numpy()
    ref, _ = ho.forward(p0, x.cpu().numpy())
    assert np.abs(probs - ref).max() < 1e-6
    assert np.array_equal(probs.argmax(1), ref.argmax(1))                 # label indices bit-exact
    loss_sum, ncorrect = hd.loss_grad(x, y).tolist()
    loss, g, nc, lsum = ho.loss_and_grad(p0, x.cpu().numpy(), y.cpu().numpy())
    assert abs(loss_sum - lsum) < 1e-4 * max(1.0, lsum) and int(ncorrect) == nc
    got = hd.grad_view().cpu().numpy()
    assert got.shape == (18507,)
    assert np.abs(got - g).max() / np.abs(g).max() < 2e-5


def test_adam_trajectory_and_determinism(dev):
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    p0 = ho.glorot_uniform_params(seed=1)
    x, y = _mk(dev, 200, seed=5)
    hd, hd2 = Head(params=p0, max_batch=256), Head(params=p0, max_batch=256)
    opt, p = ho.KerasAdam(len(p0), lr=1e-3), p0.astype(np.float64)
    losses = []
    for t in range(20):
        losses.append(hd.loss_grad(x, y).tolist()[0] / 200)
        hd.adam_step(lr=1e-3)
        hd2.loss_grad(x, y); hd2.adam_step(lr=1e-3)
        _, g, _, _ = ho.loss_and_grad(p, x.cpu().numpy(), y.cpu().numpy())
        p = opt.step(p, g)
    assert np.abs(hd.get_params() - p).max() < 2e-6
    assert np.array_equal(hd.get_params(), hd2.get_params())              # fixed-order reductions: bit-reproducible
    assert losses[-1] < losses[0] - 0.01
    assert np.abs(hd.param_view().cpu().numpy() - hd.get_params()).max() == 0


def test_data_parallel_arithmetic_on_one_gpu(dev):
    """Two half-batch gradients summed and scaled by 1/2 == the full-batch gradient (what the RCCL
    all-reduce + grad_scale = 1/world computes)."""
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    p0 = ho.glorot_uniform_params(seed=2)
    x, y = _mk(dev, 128, seed=9)
    full, a, b = (Head(params=p0, max_batch=128) for _ in range(3))
    full.loss_grad(x, y)
    a.loss_grad(x[:64], y[:64]); b.loss_grad(x[64:], y[64:])
    summed = a.grad_view() + b.grad_view()
    assert (summed * 0.5 - full.grad_view()).abs().max() / full.grad_view().abs().max() < 1e-5
    a.grad_view().copy_(summed)
    a.adam_step(lr=1e-3, grad_scale=0.5)
    full.adam_step(lr=1e-3)
    assert np.abs(a.get_params() - full.get_params()).max() < 1e-6


def test_error_contract(dev):
    from multilingual_kws_amd._lib import MkwsError
    from multilingual_kws_amd.head import Head
    hd = Head(max_batch=8, seed=0)
    x, y = _mk(dev, 9)
    with pytest.raises(MkwsError):
        hd.loss_grad(x, y)                       # batch > max_batch
    with pytest.raises(MkwsError):
        Head(hidden=64)                          # beyond the kernels' register tiling
    assert hd.forward(x[:0]).shape == (0, 3)


def test_many_heads_one_launch(dev):
    """Multi-keyword serving: N heads over one embedding batch == N separate forwards, bit for bit
    (70 heads also crosses the 64-heads-per-launch chunking)."""
    from multilingual_kws_amd.head import Head
    x, _ = _mk(dev, 37, seed=9)
    heads = [Head(max_batch=64, seed=2000 + k) for k in range(70)]
    many = Head.forward_many(heads, x)
    assert many.shape == (70, 37, 3)
    for k in (0, 1, 63, 64, 69):
        assert torch.equal(many[k], heads[k].forward(x)), k
    assert Head.forward_many(heads[:1], x[:0]).shape == (1, 0, 3)
