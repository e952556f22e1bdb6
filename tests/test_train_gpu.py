"""-m gpu: `backprop_into_embedding=True` (reference transfer_learning.py:94-112).
Training operators (mkws_op_*) against plain PyTorch-CPU fp32/fp64 references of the same op, the whole
training-mode network's gradients against oracle/efficientnet_train_oracle.py (float64 autograd, itself checked
against finite differences in tests/test_oracle_train.py), and the two-phase transfer_learn call."""
import ctypes

import numpy as np
import pytest

from tests.util_data import make_fewshot_dataset

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from multilingual_kws_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")

    class Ops:
        pass
    o = Ops()
    o.L, o.dev, o.check, o.s = L, dev, _lib.check, _lib.current_stream_ptr
    o.p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    o.t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    # scratch arena for the partial sums of the fixed-order reductions (include/mkws.h: mkws_op_set_scratch)
    o.scratch = torch.empty(4 << 20, dtype=torch.float32, device=dev)
    _lib.check(L.mkws_op_set_scratch(o.p(o.scratch), o.scratch.numel()))
    return o


@pytest.fixture(params=[0, 2], ids=["tn-staged", "tn-ring"])
def ring_tn(request, ops):
    """Weight gradients on the LDS-staged kernel (the shipped default) and on the register-ring kernel (mkws_op_set_option "gemm_ring_tn")."""
    old = ops.L.mkws_op_get_option(b"gemm_ring_tn")
    ops.check(ops.L.mkws_op_set_option(b"gemm_ring_tn", request.param))
    yield request.param
    ops.check(ops.L.mkws_op_set_option(b"gemm_ring_tn", old))


def test_training_operator_options(ops):
    L = ops.L
    assert L.mkws_op_get_option(b"gemm_ring") == 1 and L.mkws_op_get_option(b"gemm_ring_tn") == 0          # shipped defaults
    assert L.mkws_op_set_option(b"gemm_ring_tn", 3) < 0 and L.mkws_op_set_option(b"no_such_option", 1) < 0 and L.mkws_op_get_option(b"no_such_option") < 0
    for name, vals in ((b"gemm_ring", (0, 1)), (b"gemm_ring_tn", (1, 2, 0))):
        for v in vals:
            assert L.mkws_op_set_option(name, v) == 0 and L.mkws_op_get_option(name) == v


@pytest.mark.parametrize("M,N,K,ta,tb,ks", [(130, 96, 16, 0, 0, 1), (77, 24, 144, 0, 0, 1), (64, 40, 100, 0, 1, 1), (16, 96, 4000, 1, 0, 7),
                                            (240, 10, 6, 0, 0, 1), (5, 6, 240, 0, 1, 1), (2048, 1024, 8, 1, 0, 1), (33, 65, 17, 1, 1, 1),
                                            (64, 2048, 2048, 0, 0, 0), (16, 96, 32000, 1, 0, 0), (64, 300, 1000, 0, 1, 0), (100, 70, 33, 0, 0, 0),
                                            # shapes of the register-ring kernel (NN / NT with K, N % 4 == 0; round 5): two row tiles per wave
                                            # (M large), K tails of 4 / 8 / 12 columns, ragged last row / column tiles, explicit split-K
                                            (70000, 16, 32, 0, 0, 1), (4099, 144, 24, 0, 0, 1), (4099, 24, 144, 0, 1, 1), (300, 96, 2048, 0, 0, 4),
                                            (513, 112, 672, 0, 1, 0), (1000, 40, 8, 0, 0, 1), (50, 4, 4, 0, 0, 1), (6144, 480, 80, 0, 0, 1),
                                            (2048, 192, 1152, 0, 1, 0), (777, 80, 44, 0, 0, 1), (777, 44, 80, 0, 1, 1), (130, 240, 12, 0, 0, 1),
                                            # weight gradients of the big-image layers on the register-ring TN kernel (small outputs over >= 4096 rows):
                                            # row tails, ragged output tiles, one and two k-tiles per workgroup
                                            (32, 16, 70000, 1, 0, 0), (96, 24, 5003, 1, 0, 0), (24, 144, 4099, 1, 0, 0), (144, 40, 17920, 1, 0, 0),
                                            (8, 4, 4096, 1, 0, 0), (240, 240, 4500, 1, 0, 0), (40, 240, 9000, 1, 0, 1)])
def test_gemm_all_transposes(ops, M, N, K, ta, tb, ks, ring_tn):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    dA, dB = ops.t(A), ops.t(B)
    C = torch.zeros((M, N), dtype=torch.float32, device=ops.dev)
    ops.check(ops.L.mkws_op_gemm(ops.p(dA), ops.p(dB), ops.p(C), M, N, K, A.shape[1], B.shape[1], N, ta, tb, 0, ks, ops.s()))
    assert _rel(C.cpu().numpy(), ref) < 1e-5
    if ks != 1:      # split reductions (explicit, or ks = 0: chosen by the library) fold in a fixed order: bit-reproducible, and they accumulate too
        C2 = torch.full((M, N), 7.0, dtype=torch.float32, device=ops.dev)
        ops.check(ops.L.mkws_op_gemm(ops.p(dA), ops.p(dB), ops.p(C2), M, N, K, A.shape[1], B.shape[1], N, ta, tb, 0, ks, ops.s()))
        assert torch.equal(C2, C)
        ops.check(ops.L.mkws_op_gemm(ops.p(dA), ops.p(dB), ops.p(C2), M, N, K, A.shape[1], B.shape[1], N, ta, tb, 1, ks, ops.s()))
        assert _rel(C2.cpu().numpy(), 2 * ref) < 1e-5
    if ks == 1:      # accumulate
        ops.check(ops.L.mkws_op_gemm(ops.p(dA), ops.p(dB), ops.p(C), M, N, K, A.shape[1], B.shape[1], N, ta, tb, 1, 1, ops.s()))
        assert _rel(C.cpu().numpy(), 2 * ref) < 1e-5


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_gemm_on_weight_views_at_blob_offsets(ops, shift, ring_tn):
    """The trainer's weights and weight gradients are views into flat buffers at their blob offsets -- for most tensors 8 bytes off a 16-byte
    boundary.  The register-ring kernels take B (NN / NT) and C (TN) at any 4-byte alignment: float4, float2 or dword pieces of the NT fragment,
    dword loads for NN, element stores for the weight gradient.  Results must equal those on aligned copies bit for bit (same arithmetic)."""
    rng = np.random.default_rng(shift)
    M, K, N = 1000, 96, 24
    X, W, dZ = (rng.standard_normal(sh).astype(np.float32) for sh in ((M, K), (K, N), (M, N)))
    dX_ref, dW_ref, Z_ref = dZ.astype(np.float64) @ W.T.astype(np.float64), X.T.astype(np.float64) @ dZ.astype(np.float64), X.astype(np.float64) @ W.astype(np.float64)
    flat = torch.zeros(K * N + 8, dtype=torch.float32, device=ops.dev)
    gflat = torch.full((K * N + 8,), 5.0, dtype=torch.float32, device=ops.dev)
    Wv, dWv = flat[shift:shift + K * N].view(K, N), gflat[shift:shift + K * N].view(K, N)
    Wv.copy_(ops.t(W))
    assert Wv.data_ptr() % 16 == 4 * shift
    dX, dZd, Xd = ops.t(X), ops.t(dZ), ops.t(X)
    outs = {}
    for name, Wt, dWt in (("view", Wv, dWv), ("aligned", ops.t(W), torch.full((K, N), 5.0, dtype=torch.float32, device=ops.dev))):
        Z = torch.empty((M, N), dtype=torch.float32, device=ops.dev)
        dXo = torch.empty((M, K), dtype=torch.float32, device=ops.dev)
        ops.check(ops.L.mkws_op_gemm(ops.p(Xd), ops.p(Wt), ops.p(Z), M, N, K, K, N, N, 0, 0, 0, 1, ops.s()))               # forward: NN
        ops.check(ops.L.mkws_op_gemm(ops.p(dZd), ops.p(Wt), ops.p(dXo), M, K, N, N, N, K, 0, 1, 0, 1, ops.s()))            # input gradient: NT
        ops.check(ops.L.mkws_op_gemm(ops.p(Xd), ops.p(dZd), ops.p(dWt), K, N, M, K, N, N, 1, 0, 0, 0, ops.s()))            # weight gradient: TN (split, folded)
        dW1 = dWt.clone()
        ops.check(ops.L.mkws_op_gemm(ops.p(Xd), ops.p(dZd), ops.p(dWt), K, N, M, K, N, N, 1, 0, 1, 1, ops.s()))            # TN, unsplit, accumulating (direct stores)
        outs[name] = (Z, dXo, dW1, dWt.clone())
    for a, b in zip(outs["view"], outs["aligned"]):
        assert torch.equal(a, b)
    Z, dXo, dW1, dW2 = outs["view"]
    assert _rel(Z.cpu().numpy(), Z_ref) < 1e-5 and _rel(dXo.cpu().numpy(), dX_ref) < 1e-5 and _rel(dW1.cpu().numpy(), dW_ref) < 1e-5
    assert _rel(dW2.cpu().numpy(), 2 * dW_ref) < 1e-5
    assert float(gflat[:shift].min()) == 5.0 and float(gflat[shift + K * N:].min()) == 5.0 and float(gflat[shift + K * N:].max()) == 5.0     # nothing stored around the view


@pytest.mark.parametrize("B,HW,C,act", [(6, 35, 40, 0), (5, 130, 24, 1)])
def test_batchnorm_fused_residual_and_gradient_assembly(ops, B, HW, C, act):
    """mkws_op_bn_train_fwd_res / mkws_op_bn_act_bwd_ex = the unfused operator sequences (row_scale_add, add_bcast + bn_act_bwd)."""
    ops.check(ops.L.mkws_op_set_scratch(ops.p(ops.scratch), ops.scratch.numel()))
    rng = np.random.default_rng(B * HW + C)
    M = B * HW
    Z, X = ops.t(rng.standard_normal((M, C))), ops.t(rng.standard_normal((M, C)))
    g, b = ops.t(rng.uniform(0.5, 1.5, C)), ops.t(0.1 * rng.standard_normal(C))
    keep = ops.t((rng.random(B) > 0.3) / 0.7)
    new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=ops.dev)

    def stats():
        return ops.t(np.zeros(C)), ops.t(np.ones(C)), new(C), new(C)
    mm, mv, mean, var = stats()
    A = new(M, C)
    ops.check(ops.L.mkws_op_bn_train_fwd(ops.p(Z), M, C, ops.p(g), ops.p(b), 1e-3, act, 0.99, ops.p(mm), ops.p(mv), ops.p(mean), ops.p(var), ops.p(A), ops.s()))
    ref = new(M, C)
    ops.check(ops.L.mkws_op_row_scale_add(ops.p(A), ops.p(keep), ops.p(X), ops.p(ref), B, HW * C, ops.s()))
    mm2, mv2, mean2, var2 = stats()
    out = new(M, C)
    ops.check(ops.L.mkws_op_bn_train_fwd_res(ops.p(Z), M, C, ops.p(g), ops.p(b), 1e-3, act, 0.99, ops.p(mm2), ops.p(mv2), ops.p(mean2), ops.p(var2), ops.p(out),
                                             ops.p(X), ops.p(keep), HW, ops.s()))
    assert torch.equal(mean2, mean) and torch.equal(var2, var) and torch.equal(mm2, mm) and torch.equal(mv2, mv)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    # backward: incoming gradient = src * keep[row // HW] + bcast[row // HW] / HW, assembled inside the first launch
    src, bc = ops.t(rng.standard_normal((M, C))), ops.t(rng.standard_normal((B, C)))
    dA = new(M, C)
    ops.check(ops.L.mkws_op_row_scale_add(ops.p(src), ops.p(keep), None, ops.p(dA), B, HW * C, ops.s()))
    ops.check(ops.L.mkws_op_add_bcast(ops.p(dA), ops.p(bc), 1.0 / HW, B, HW, C, ops.s()))
    gg, gb, scr = new(C), new(C), new(2 * C)
    ops.check(ops.L.mkws_op_bn_act_bwd(ops.p(Z), ops.p(mean), ops.p(var), ops.p(g), ops.p(b), 1e-3, act, ops.p(dA), ops.p(gg), ops.p(gb), ops.p(scr), M, C, ops.s()))
    keep_src = src.clone()
    d2, gg2, gb2 = new(M, C), new(C), new(C)
    ops.check(ops.L.mkws_op_bn_act_bwd_ex(ops.p(Z), ops.p(mean), ops.p(var), ops.p(g), ops.p(b), 1e-3, act, ops.p(d2), ops.p(src), ops.p(keep), ops.p(bc), 1.0 / HW, HW,
                                          ops.p(gg2), ops.p(gb2), M, C, ops.s()))
    assert torch.equal(src, keep_src)                                   # the source gradient stays intact (the shortcut still needs it)
    for got, want in ((d2, dA), (gg2, gg), (gb2, gb)):
        assert _rel(got.cpu().numpy(), want.cpu().numpy()) < 1e-5
    # neither a source nor a broadcast term: refused
    assert ops.L.mkws_op_bn_act_bwd_ex(ops.p(Z), ops.p(mean), ops.p(var), ops.p(g), ops.p(b), 1e-3, act, ops.p(d2), None, None, None, 0.0, HW, ops.p(gg2), ops.p(gb2), M, C,
                                       ops.s()) < 0


def test_deferred_folds_are_bit_identical(ops):
    """mkws_op_fold_defer: queued second stages (weight-gradient GEMM, bias / depthwise / stem gradient folds) = the immediate ones, bit for bit,
    also when the queue overflows (more than 24 entries) and when a flush comes in between."""
    ops.check(ops.L.mkws_op_set_scratch(ops.p(ops.scratch), ops.scratch.numel()))
    rng = np.random.default_rng(11)
    new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=ops.dev)
    M, K, N = 4000, 24, 96
    X, dZ = ops.t(rng.standard_normal((M, K))), ops.t(rng.standard_normal((M, N)))
    Zb, bias = ops.t(rng.standard_normal((M, N))), ops.t(rng.standard_normal(N))
    B, H, W, C, k = 8, 13, 10, 48, 3
    Xd, Wd, dZd = ops.t(rng.standard_normal((B * H * W, C))), ops.t(rng.standard_normal((k * k, C))), ops.t(rng.standard_normal((B * H * W, C)))

    def run(defer, rounds):
        outs = []
        ops.check(ops.L.mkws_op_fold_defer(1 if defer else 0, ops.s()))
        for r in range(rounds):
            dW, db, dWd = new(K, N), new(N), new(k * k, C)
            dA = dZ.clone()
            ops.check(ops.L.mkws_op_gemm(ops.p(X), ops.p(dZ), ops.p(dW), K, N, M, K, N, N, 1, 0, 0, 0, ops.s()))
            ops.check(ops.L.mkws_op_bias_act_bwd(ops.p(Zb), ops.p(bias), 1, ops.p(dA), ops.p(db), M, N, ops.s()))
            ops.check(ops.L.mkws_op_dwconv_bwd(ops.p(Xd), ops.p(Wd), ops.p(dZd), None, ops.p(dWd), B, H, W, C, k, 1, 1, 1, H, W, ops.s()))
            if defer and r == 3:
                ops.check(ops.L.mkws_op_fold_flush(ops.s()))
            outs += [dW, db, dWd]
        ops.check(ops.L.mkws_op_fold_defer(0, ops.s()))
        torch.cuda.synchronize()
        return outs
    ref = run(False, 12)
    got = run(True, 12)                                                # 36 folds: the 24-entry queue overflows once, plus one explicit flush
    assert all(torch.equal(a, b) for a, b in zip(got, ref))
    assert all(torch.isfinite(t).all() for t in got)


@pytest.mark.parametrize("M,C,act", [(1000, 32, 1), (48, 240, 1), (96, 24, 0), (4000, 96, 1)])
def test_batchnorm_train_forward_backward(ops, M, C, act):
    rng = np.random.default_rng(C)
    Z = (rng.standard_normal((M, C)) * rng.uniform(0.2, 3, C) + rng.standard_normal(C)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, C).astype(np.float32), (0.1 * rng.standard_normal(C)).astype(np.float32)
    dA = rng.standard_normal((M, C)).astype(np.float32)
    z = torch.tensor(Z, dtype=torch.float64, requires_grad=True)
    g, b = torch.tensor(gamma, dtype=torch.float64, requires_grad=True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mean, var = z.mean(0), z.var(0, unbiased=False)
    y = g * (z - mean) / torch.sqrt(var + 1e-3) + b
    a = y * torch.sigmoid(y) if act == 1 else y
    (a * torch.tensor(dA, dtype=torch.float64)).sum().backward()
    dZ, dm, dv = ops.t(Z), torch.empty(C, device=ops.dev), torch.empty(C, device=ops.dev)
    ops.check(ops.L.mkws_op_bn_stats(ops.p(dZ), M, C, ops.p(dm), ops.p(dv), ops.s()))
    assert _rel(dm.cpu().numpy(), mean.detach().numpy()) < 1e-5 and _rel(dv.cpu().numpy(), var.detach().numpy()) < 1e-5
    dg, db, A = ops.t(gamma), ops.t(beta), torch.empty((M, C), device=ops.dev)
    ops.check(ops.L.mkws_op_bn_act_fwd(ops.p(dZ), ops.p(dm), ops.p(dv), ops.p(dg), ops.p(db), 1e-3, act, ops.p(A), M, C, ops.s()))
    assert _rel(A.cpu().numpy(), a.detach().numpy()) < 1e-5
    d, gg, gb, scr = ops.t(dA), torch.empty(C, device=ops.dev), torch.empty(C, device=ops.dev), torch.empty(2 * C, device=ops.dev)
    ops.check(ops.L.mkws_op_bn_act_bwd(ops.p(dZ), ops.p(dm), ops.p(dv), ops.p(dg), ops.p(db), 1e-3, act, ops.p(d), ops.p(gg), ops.p(gb), ops.p(scr), M, C, ops.s()))
    assert _rel(d.cpu().numpy(), z.grad.numpy()) < 2e-4
    assert _rel(gg.cpu().numpy(), g.grad.numpy()) < 2e-4 and _rel(gb.cpu().numpy(), b.grad.numpy()) < 2e-4
    mm, mv = ops.t(np.zeros(C)), ops.t(np.ones(C))
    ops.check(ops.L.mkws_op_bn_update_moving(ops.p(mm), ops.p(mv), ops.p(dm), ops.p(dv), 0.99, M, C, ops.s()))
    assert _rel(mm.cpu().numpy(), 0.01 * mean.detach().numpy()) < 1e-5
    assert _rel(mv.cpu().numpy(), 0.99 + 0.01 * var.detach().numpy() * M / (M - 1)) < 1e-5
    # the fused training forward (statistics, then moving update + normalise / activate: two launches) gives the same statistics bit for bit
    mm2, mv2 = ops.t(np.zeros(C)), ops.t(np.ones(C))
    dm2, dv2, A2 = torch.empty(C, device=ops.dev), torch.empty(C, device=ops.dev), torch.empty((M, C), device=ops.dev)
    ops.check(ops.L.mkws_op_bn_train_fwd(ops.p(dZ), M, C, ops.p(dg), ops.p(db), 1e-3, act, 0.99, ops.p(mm2), ops.p(mv2), ops.p(dm2), ops.p(dv2), ops.p(A2), ops.s()))
    assert torch.equal(dm2, dm) and torch.equal(dv2, dv) and torch.equal(mm2, mm) and torch.equal(mv2, mv) and torch.allclose(A2, A, rtol=1e-6, atol=1e-6)
    # fixed-order reductions: the backward pass repeats bit for bit
    d2, gg2, gb2 = ops.t(dA), torch.empty(C, device=ops.dev), torch.empty(C, device=ops.dev)
    ops.check(ops.L.mkws_op_bn_act_bwd(ops.p(dZ), ops.p(dm), ops.p(dv), ops.p(dg), ops.p(db), 1e-3, act, ops.p(d2), ops.p(gg2), ops.p(gb2), ops.p(scr), M, C, ops.s()))
    assert torch.equal(d2, d) and torch.equal(gg2, gg) and torch.equal(gb2, gb)
    # without a scratch arena the op fails loudly instead of falling back to atomics
    ops.check(ops.L.mkws_op_set_scratch(None, 0))
    assert ops.L.mkws_op_bn_stats(ops.p(dZ), M, C, ops.p(dm), ops.p(dv), ops.s()) < 0 and b"scratch" in ops.L.mkws_last_error()
    ops.check(ops.L.mkws_op_set_scratch(ops.p(ops.scratch), ops.scratch.numel()))


@pytest.mark.parametrize("H,W,C,k,s", [(25, 20, 96, 3, 2), (13, 10, 144, 5, 2), (7, 5, 240, 5, 1), (4, 3, 480, 3, 1), (4, 3, 672, 5, 2), (2, 2, 1152, 5, 1)])
def test_depthwise_forward_backward(ops, H, W, C, k, s):
    from oracle.efficientnet_oracle import correct_pad
    rng = np.random.default_rng(H * C)
    B = 3
    X = rng.standard_normal((B, H, W, C)).astype(np.float32)
    Wt = rng.standard_normal((k, k, C)).astype(np.float32)
    if s == 2:
        (pt, pb), (pl, pr) = correct_pad(H, W, k)
    else:
        pt = pb = pl = pr = k // 2
    x = torch.tensor(X, dtype=torch.float64).permute(0, 3, 1, 2).requires_grad_(True)
    w = torch.tensor(Wt, dtype=torch.float64).permute(2, 0, 1)[:, None].requires_grad_(True)
    z = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, stride=s, groups=C)
    Ho, Wo = z.shape[2], z.shape[3]
    dZ = rng.standard_normal((B, Ho, Wo, C)).astype(np.float32)
    (z * torch.tensor(dZ, dtype=torch.float64).permute(0, 3, 1, 2)).sum().backward()
    dx, dw, dz, Z = ops.t(X), ops.t(Wt), ops.t(dZ), torch.empty((B, Ho, Wo, C), device=ops.dev)
    ops.check(ops.L.mkws_op_dwconv_fwd(ops.p(dx), ops.p(dw), ops.p(Z), B, H, W, C, k, s, pt, pl, Ho, Wo, ops.s()))
    assert _rel(Z.cpu().numpy(), z.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    gX, gW = torch.empty_like(dx), torch.empty_like(dw)
    ops.check(ops.L.mkws_op_dwconv_bwd(ops.p(dx), ops.p(dw), ops.p(dz), ops.p(gX), ops.p(gW), B, H, W, C, k, s, pt, pl, Ho, Wo, ops.s()))
    assert _rel(gX.cpu().numpy(), x.grad.permute(0, 2, 3, 1).numpy()) < 1e-5
    assert _rel(gW.cpu().numpy(), w.grad[:, 0].permute(1, 2, 0).numpy()) < 1e-4


def test_stem_se_and_elementwise_operators(ops):
    rng = np.random.default_rng(0)
    B = 4
    spec = (rng.integers(0, 670, size=(B, 49, 40)) * (10 / 256)).astype(np.float32)
    Wt = rng.standard_normal((3, 3, 1, 32)).astype(np.float32)
    x = (torch.tensor(spec, dtype=torch.float64)[:, None] / 255.0 - 0.1) / 0.7
    w = torch.tensor(Wt, dtype=torch.float64).permute(3, 2, 0, 1).requires_grad_(True)
    z = F.conv2d(F.pad(x, (0, 1, 1, 1)), w, stride=2)
    dZ = rng.standard_normal((B, 25, 20, 32)).astype(np.float32)
    (z * torch.tensor(dZ, dtype=torch.float64).permute(0, 3, 1, 2)).sum().backward()
    ds, dw, Z = ops.t(spec), ops.t(Wt), torch.empty((B, 25, 20, 32), device=ops.dev)
    ops.check(ops.L.mkws_op_stem_fwd(ops.p(ds), ops.p(dw), 0.1, 0.7, ops.p(Z), B, ops.s()))
    assert _rel(Z.cpu().numpy(), z.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    gW, ddZ = torch.empty((3, 3, 1, 32), device=ops.dev), ops.t(dZ)
    ops.check(ops.L.mkws_op_stem_bwd_weight(ops.p(ds), ops.p(ddZ), 0.1, 0.7, ops.p(gW), B, ops.s()))
    assert _rel(gW.cpu().numpy(), w.grad.permute(2, 3, 1, 0).numpy()) < 1e-4
    # SE pieces
    HW, C = 12, 80
    A, g, dO = rng.standard_normal((B, HW, C)).astype(np.float32), rng.uniform(0, 1, (B, C)).astype(np.float32), rng.standard_normal((B, HW, C)).astype(np.float32)
    dA_, dg_, dOut = ops.t(A), ops.t(g), ops.t(dO)
    mean, out = torch.empty((B, C), device=ops.dev), torch.empty((B, HW, C), device=ops.dev)
    ops.check(ops.L.mkws_op_pool_hw(ops.p(dA_), ops.p(mean), B, HW, C, ops.s()))
    assert _rel(mean.cpu().numpy(), A.mean(1)) < 1e-6
    ops.check(ops.L.mkws_op_scale_channels(ops.p(dA_), ops.p(dg_), ops.p(out), B, HW, C, ops.s()))
    assert _rel(out.cpu().numpy(), A * g[:, None]) < 1e-6
    gA, gg = torch.empty((B, HW, C), device=ops.dev), torch.empty((B, C), device=ops.dev)
    ops.check(ops.L.mkws_op_se_bwd(ops.p(dA_), ops.p(dg_), ops.p(dOut), ops.p(gA), ops.p(gg), B, HW, C, ops.s()))
    assert _rel(gA.cpu().numpy(), dO * g[:, None]) < 1e-6 and _rel(gg.cpu().numpy(), (dO * A).sum(1)) < 1e-5
    ops.check(ops.L.mkws_op_add_bcast(ops.p(gA), ops.p(dg_), 0.25, B, HW, C, ops.s()))
    assert _rel(gA.cpu().numpy(), dO * g[:, None] + 0.25 * g[:, None]) < 1e-6
    # bias + activation and its backward, all four activations
    M, N = 37, 50
    Zb, bias, dAb = rng.standard_normal((M, N)).astype(np.float32), rng.standard_normal(N).astype(np.float32), rng.standard_normal((M, N)).astype(np.float32)
    fns = {1: lambda y: y * torch.sigmoid(y), 2: torch.relu, 3: torch.selu, 4: torch.sigmoid}
    for act, fn in fns.items():
        zz = torch.tensor(Zb, dtype=torch.float64, requires_grad=True)
        bb = torch.tensor(bias, dtype=torch.float64, requires_grad=True)
        a = fn(zz + bb)
        (a * torch.tensor(dAb, dtype=torch.float64)).sum().backward()
        Ab, d, gb = torch.empty((M, N), device=ops.dev), ops.t(dAb), torch.empty(N, device=ops.dev)
        dZb, dbias = ops.t(Zb), ops.t(bias)                 # (kept alive: a temporary's memory would be recycled under the kernel)
        ops.check(ops.L.mkws_op_bias_act_fwd(ops.p(dZb), ops.p(dbias), act, ops.p(Ab), M, N, ops.s()))
        assert _rel(Ab.cpu().numpy(), a.detach().numpy()) < 1e-6, act
        ops.check(ops.L.mkws_op_bias_act_bwd(ops.p(dZb), ops.p(dbias), act, ops.p(d), ops.p(gb), M, N, ops.s()))
        assert _rel(d.cpu().numpy(), zz.grad.numpy()) < 1e-5 and _rel(gb.cpu().numpy(), bb.grad.numpy()) < 1e-5, act
    # dense / SE forward with the bias + activation epilogue fused into the GEMM: unsplit (K = 48) and split-reduction (K = 2048) forms
    for Md, Nd, Kd, act in ((37, 50, 48, 1), (64, 72, 2048, 2), (5, 20, 1152, 4)):
        Xd, Wd = rng.standard_normal((Md, Kd)).astype(np.float32), (rng.standard_normal((Kd, Nd)) / np.sqrt(Kd)).astype(np.float32)
        bd = rng.standard_normal(Nd).astype(np.float32)
        zref = Xd.astype(np.float64) @ Wd.astype(np.float64)
        aref = fns[act](torch.tensor(zref + bd)).numpy()
        dX_, dW_, db_ = ops.t(Xd), ops.t(Wd), ops.t(bd)
        Zd, Ad = torch.empty((Md, Nd), device=ops.dev), torch.empty((Md, Nd), device=ops.dev)
        ops.check(ops.L.mkws_op_dense_fwd(ops.p(dX_), ops.p(dW_), ops.p(db_), act, ops.p(Zd), ops.p(Ad), Md, Nd, Kd, ops.s()))
        assert _rel(Zd.cpu().numpy(), zref) < 1e-5 and _rel(Ad.cpu().numpy(), aref) < 1e-5, (Md, Nd, Kd)
    # drop-connect + residual, axpy, Adam
    a, c, sc = ops.t(A), ops.t(dO), ops.t(np.array([0.0, 1.25, 1.25, 0.0]))
    out = torch.empty((B, HW, C), device=ops.dev)
    ops.check(ops.L.mkws_op_row_scale_add(ops.p(a), ops.p(sc), ops.p(c), ops.p(out), B, HW * C, ops.s()))
    assert _rel(out.cpu().numpy(), A * np.array([0, 1.25, 1.25, 0], np.float32)[:, None, None] + dO) < 1e-6
    ops.check(ops.L.mkws_op_axpy(ops.p(out), ops.p(a), -2.0, out.numel(), ops.s()))
    assert _rel(out.cpu().numpy(), A * np.array([0, 1.25, 1.25, 0], np.float32)[:, None, None] + dO - 2 * A) < 1e-5
    from oracle import head_oracle as ho
    n = 1000
    p0, gr = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    P, Gd, m, v = ops.t(p0), ops.t(gr), ops.t(np.zeros(n)), ops.t(np.zeros(n))
    opt, pr = ho.KerasAdam(n, lr=1e-3), p0.astype(np.float64)
    for t in range(1, 4):
        ops.check(ops.L.mkws_op_adam(ops.p(P), ops.p(Gd), ops.p(m), ops.p(v), n, 1e-3, 0.9, 0.999, 1e-7, t, 0.5, ops.s()))
        pr = opt.step(pr, 0.5 * gr.astype(np.float64))
    assert np.abs(P.cpu().numpy() - pr).max() < 1e-6
    # the graph-replayable form (step index on the device) walks the same trajectory
    P2, m2, v2 = ops.t(p0), ops.t(np.zeros(n)), ops.t(np.zeros(n))
    step = torch.zeros(1, dtype=torch.int32, device=ops.dev)
    for t in range(1, 4):
        ops.check(ops.L.mkws_op_step_inc(ops.p(step), ops.s()))
        ops.check(ops.L.mkws_op_adam_dev(ops.p(P2), ops.p(Gd), ops.p(m2), ops.p(v2), n, 1e-3, 0.9, 0.999, 1e-7, ops.p(step), 0.5, ops.s()))
    assert int(step.item()) == 3 and np.abs(P2.cpu().numpy() - P.cpu().numpy()).max() < 1e-7


@pytest.mark.parametrize("M,K,N,act,res", [(768, 80, 480, 1, False), (2240, 144, 40, 0, True), (100, 16, 96, 1, False), (256, 1152, 320, 0, False),
                                           (12800, 16, 96, 1, False)])
def test_conv_batchnorm_as_one_operator(ops, M, K, N, act, res):
    """mkws_op_conv_bn_fwd == mkws_op_gemm + mkws_op_bn_train_fwd_res.  Unsplit GEMMs with <= 160 row tiles leave the BN chunk statistics
    in their epilogue (other chunking than the statistics kernel: statistics to round-off, not to the bit); split GEMMs (K = 1152) and
    large M (200 row tiles) take the sequence itself: bit-identical."""
    rng = np.random.default_rng(M + N)
    X, W = ops.t(rng.standard_normal((M, K))), ops.t(rng.standard_normal((K, N)) / np.sqrt(K))
    g, b = ops.t(rng.uniform(0.5, 1.5, N)), ops.t(rng.standard_normal(N) * 0.1)
    group = 4
    R = ops.t(rng.standard_normal((M, N))) if res else None
    rs = ops.t(rng.uniform(0, 1.3, M // group)) if res else None
    e = lambda *shape: torch.full(shape, float("nan"), device=ops.dev)
    out = []
    for fused in (False, True):
        mm, mv = ops.t(np.zeros(N)), ops.t(np.ones(N))
        Z, mean, var, A = e(M, N), e(N), e(N), e(M, N)
        if fused:
            ops.check(ops.L.mkws_op_conv_bn_fwd(ops.p(X), ops.p(W), ops.p(Z), M, N, K, ops.p(g), ops.p(b), 1e-3, act, 0.99, ops.p(mm), ops.p(mv), ops.p(mean),
                                                ops.p(var), ops.p(A), ops.p(R), ops.p(rs), group, ops.s()))
        else:
            ops.check(ops.L.mkws_op_gemm(ops.p(X), ops.p(W), ops.p(Z), M, N, K, K, N, N, 0, 0, 0, 0, ops.s()))
            ops.check(ops.L.mkws_op_bn_train_fwd_res(ops.p(Z), M, N, ops.p(g), ops.p(b), 1e-3, act, 0.99, ops.p(mm), ops.p(mv), ops.p(mean), ops.p(var), ops.p(A),
                                                     ops.p(R), ops.p(rs), group, ops.s()))
        out.append([t.cpu().numpy() for t in (Z, mean, var, A, mm, mv)])
    (Z0, m0, v0, A0, mm0, mv0), (Z1, m1, v1, A1, mm1, mv1) = out
    assert np.array_equal(Z0, Z1)
    if K >= 512 or M > 160 * 64:
        assert all(np.array_equal(a, b_) for a, b_ in zip(out[0], out[1]))
    z64 = Z0.astype(np.float64)
    assert np.abs(m1 - z64.mean(0)).max() < 1e-5 and _rel(v1, z64.var(0)) < 1e-5
    assert _rel(m1, m0) < 1e-5 and _rel(v1, v0) < 1e-5 and _rel(A1, A0) < 1e-4 and _rel(mm1, mm0) < 1e-5 and _rel(mv1, mv0) < 1e-5


@pytest.mark.parametrize("B,H,W,C,k,s", [(3, 25, 20, 96, 3, 2), (5, 7, 5, 240, 5, 1), (70, 2, 2, 1152, 3, 1), (9, 13, 10, 144, 5, 2)])
def test_depthwise_batchnorm_as_one_operator(ops, B, H, W, C, k, s):
    """mkws_op_dwconv_bn_fwd == mkws_op_dwconv_fwd + mkws_op_bn_train_fwd: same convolution bits, statistics to round-off."""
    rng = np.random.default_rng(B + C)
    if s == 2:
        pt, pl = k // 2 - (1 - H % 2), k // 2 - (1 - W % 2)
        Ho, Wo = (H + pt + k // 2 - k) // 2 + 1, (W + pl + k // 2 - k) // 2 + 1
    else:
        pt, pl, Ho, Wo = k // 2, k // 2, H, W
    M = B * Ho * Wo
    X, Wd = ops.t(rng.standard_normal((B, H, W, C))), ops.t(rng.standard_normal((k, k, C)) / k)
    g, b = ops.t(rng.uniform(0.5, 1.5, C)), ops.t(rng.standard_normal(C) * 0.1)
    e = lambda *shape: torch.full(shape, float("nan"), device=ops.dev)
    out = []
    for fused in (False, True):
        mm, mv = ops.t(np.zeros(C)), ops.t(np.ones(C))
        Z, mean, var, A = e(M, C), e(C), e(C), e(M, C)
        if fused:
            ops.check(ops.L.mkws_op_dwconv_bn_fwd(ops.p(X), ops.p(Wd), ops.p(Z), B, H, W, C, k, s, pt, pl, Ho, Wo, ops.p(g), ops.p(b), 1e-3, 1, 0.99, ops.p(mm),
                                                  ops.p(mv), ops.p(mean), ops.p(var), ops.p(A), ops.s()))
        else:
            ops.check(ops.L.mkws_op_dwconv_fwd(ops.p(X), ops.p(Wd), ops.p(Z), B, H, W, C, k, s, pt, pl, Ho, Wo, ops.s()))
            ops.check(ops.L.mkws_op_bn_train_fwd(ops.p(Z), M, C, ops.p(g), ops.p(b), 1e-3, 1, 0.99, ops.p(mm), ops.p(mv), ops.p(mean), ops.p(var), ops.p(A), ops.s()))
        out.append([t.cpu().numpy() for t in (Z, mean, var, A, mm, mv)])
    (Z0, m0, v0, A0, mm0, mv0), (Z1, m1, v1, A1, mm1, mv1) = out
    assert np.array_equal(Z0, Z1)
    z64 = Z0.astype(np.float64)
    assert np.abs(m1 - z64.mean(0)).max() < 1e-5 and _rel(v1, z64.var(0)) < 1e-5
    assert _rel(A1, A0) < 1e-4 and _rel(mm1, mm0) < 1e-5 and _rel(mv1, mv0) < 1e-5


@pytest.mark.parametrize("B,HW,C,se", [(5, 500, 32, 8), (3, 130, 96, 4), (6, 35, 240, 10), (4, 12, 672, 28), (70, 4, 1152, 48)])
def test_fused_squeeze_excite_forward_backward(ops, B, HW, C, se):
    """mkws_op_se_fwd / mkws_op_se_bwd_fused (two launches forward, three backward) against float64 autograd of
    pool -> dense(swish) -> dense(sigmoid) -> multiply; every stored intermediate and all six gradients."""
    rng = np.random.default_rng(B * 1000 + C)
    A = rng.standard_normal((B, HW, C)).astype(np.float32)
    Wr, br = (rng.standard_normal((C, se)) / np.sqrt(C)).astype(np.float32), rng.standard_normal(se).astype(np.float32)
    We, be = (rng.standard_normal((se, C)) / np.sqrt(se)).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    dO = rng.standard_normal((B, HW, C)).astype(np.float32)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    a, wr, b_r, we, b_e = t64(A), t64(Wr), t64(br), t64(We), t64(be)
    mean = a.mean(1)
    mean.retain_grad()
    yr = mean @ wr + b_r
    r = yr * torch.sigmoid(yr)
    g = torch.sigmoid(r @ we + b_e)
    out = a * g[:, None]
    (out * torch.tensor(dO, dtype=torch.float64)).sum().backward()
    dA_, dWr, dbr, dWe, dbe, ddO = ops.t(A), ops.t(Wr), ops.t(br), ops.t(We), ops.t(be), ops.t(dO)
    e = lambda *shape: torch.full(shape, float("nan"), device=ops.dev)
    Mn, Yr, R, G, Out = e(B, C), e(B, se), e(B, se), e(B, C), e(B, HW, C)
    work = e(B, (C + 127) // 128 * se)
    ops.check(ops.L.mkws_op_se_fwd(ops.p(dA_), ops.p(dWr), ops.p(dbr), ops.p(dWe), ops.p(dbe), ops.p(Mn), ops.p(Yr), ops.p(R), ops.p(G), ops.p(Out),
                                   ops.p(work), B, HW, C, se, ops.s()))
    for got, ref, name in ((Mn, mean, "mean"), (Yr, yr, "Yr"), (R, r, "R"), (G, g, "G"), (Out, out, "out")):
        assert _rel(got.cpu().numpy(), ref.detach().numpy()) < 2e-6, name
    gA, gmean, gYg, gYr = e(B, HW, C), e(B, C), e(B, C), e(B, se)
    gWr, gbr, gWe, gbe = e(C, se), e(se), e(se, C), e(C)
    ops.check(ops.L.mkws_op_se_bwd_fused(ops.p(dA_), ops.p(G), ops.p(ddO), ops.p(Mn), ops.p(Yr), ops.p(R), ops.p(dWr), ops.p(dWe), ops.p(gA), ops.p(gmean),
                                         ops.p(gYg), ops.p(gYr), ops.p(gWr), ops.p(gbr), ops.p(gWe), ops.p(gbe), ops.p(work), B, HW, C, se, ops.s()))
    # dA is the multiply's direct path only; the squeeze's path (dmean / HW on every pixel) is added by the BatchNorm backward that follows
    direct = dO.astype(np.float64) * g.detach().numpy()[:, None]
    assert _rel(gA.cpu().numpy(), direct) < 2e-6
    assert _rel(gmean.cpu().numpy(), mean.grad.numpy()) < 2e-5
    assert _rel((gA.cpu().numpy().astype(np.float64) + gmean.cpu().numpy()[:, None] / HW), a.grad.numpy()) < 2e-5
    for got, ref, name in ((gWr, wr.grad, "dWr"), (gbr, b_r.grad, "dbr"), (gWe, we.grad, "dWe"), (gbe, b_e.grad, "dbe")):
        assert _rel(got.cpu().numpy(), ref.numpy()) < 2e-5, name
    # fixed-order sums: a second call is bit-identical
    gWr2, gbr2, gWe2, gbe2, gA2, gm2 = e(C, se), e(se), e(se, C), e(C), e(B, HW, C), e(B, C)
    ops.check(ops.L.mkws_op_se_bwd_fused(ops.p(dA_), ops.p(G), ops.p(ddO), ops.p(Mn), ops.p(Yr), ops.p(R), ops.p(dWr), ops.p(dWe), ops.p(gA2), ops.p(gm2),
                                         ops.p(gYg), ops.p(gYr), ops.p(gWr2), ops.p(gbr2), ops.p(gWe2), ops.p(gbe2), ops.p(work), B, HW, C, se, ops.s()))
    assert torch.equal(gWr2, gWr) and torch.equal(gWe2, gWe) and torch.equal(gbr2, gbr) and torch.equal(gbe2, gbe) and torch.equal(gm2, gmean)


def test_training_mode_gradients_match_the_oracle():
    """Every trainable tensor's gradient (d sum(emb . proj)) in training mode (batch statistics, drop-connect) against the
    float64 autograd oracle; tolerance 1e-3 of the tensor's largest gradient."""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
    from oracle.efficientnet_train_oracle import TrainableEmbeddingOracle
    blob = weights.synthetic_blob()
    rng = np.random.default_rng(3)
    B = 6
    spec = (rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256))
    proj = rng.standard_normal(1024)
    masks = {"2b": np.array([1, 1, 0, 1, 1, 1], bool), "4c": np.array([1, 0, 1, 1, 0, 1], bool), "6d": np.array([0, 1, 1, 1, 1, 1], bool)}
    o = TrainableEmbeddingOracle(blob)
    o.zero_grad()
    ref_emb = o.forward(spec, training=True, drop_masks={k: torch.tensor(v) for k, v in masks.items()})
    (ref_emb @ torch.from_numpy(proj)).sum().backward()
    ref = o.grads()
    tr = EmbeddingTrainer(blob)
    emb = tr.forward_train(torch.from_numpy(spec).cuda(), masks)
    assert _rel(emb.cpu().numpy(), ref_emb.detach().numpy()) < 1e-4
    tr.backward(torch.from_numpy(np.tile(proj.astype(np.float32), (B, 1))).cuda())
    got = tr.named_grads()
    # A beta (or bias-like shift) that feeds a 1x1 conv followed by a batch-statistics BN has an exactly-zero gradient
    # (block1a_project_bn/beta, ...): the oracle returns 1e-17 there, fp32 1e-9, so tensors are compared relative to
    # their own largest gradient with an absolute floor of 1e-6 of the network's largest one.
    gmax = max(float(np.abs(g).max()) for g in ref.values())
    worst = {}
    for name, g in ref.items():
        err = float(np.abs(got[name].reshape(g.shape).astype(np.float64) - g).max())
        worst[name] = err / max(float(np.abs(g).max()), 1e-3 * gmax)
        assert err <= 1e-3 * float(np.abs(g).max()) + 1e-6 * gmax, (name, err, float(np.abs(g).max()), gmax)
    assert len(worst) == len(o.trainable)
    print("largest per-tensor gradient error (relative):", max(worst.values()), "over", len(worst), "tensors; gmax", gmax)
    for name in ("normalization/mean", "stem_bn/moving_mean", "top_bn/moving_variance"):
        assert not got[name].any()                                         # non-trainable: zero gradient
    # moving statistics after the training-mode forward (momentum 0.99, Bessel-corrected variance)
    newp = tr.blob()
    for name, val in o.new_moving.items():
        t = tr.tensors[name]
        assert _rel(newp[t["offset"]:t["offset"] + t["count"]], val.detach().numpy()) < 1e-5, name
    # one Adam step moves every trainable tensor by ~lr and leaves the rest alone
    before = tr.blob()
    tr.adam_step(lr=1e-3)
    after = tr.blob()
    t = tr.tensors["dense_2/kernel"]
    d = np.abs(after - before)[t["offset"]:t["offset"] + t["count"]]
    assert 0.5e-3 < d.max() < 1.01e-3
    t = tr.tensors["block5a_bn/moving_mean"]
    assert np.array_equal(after[t["offset"]:t["offset"] + t["count"]], before[t["offset"]:t["offset"] + t["count"]])


def test_head_input_gradient():
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    rng = np.random.default_rng(0)
    p0 = ho.glorot_uniform_params(seed=4)
    x = (rng.standard_normal((40, 1024)) * 0.3).astype(np.float32)
    y = rng.integers(0, 3, 40)
    hd = Head(params=p0, max_batch=64)
    hd.loss_grad(torch.from_numpy(x).cuda(), torch.from_numpy(y.astype(np.int32)).cuda())
    dx = hd.input_grad(40).cpu().numpy()
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    W1, b1, W2, b2 = [torch.tensor(a, dtype=torch.float64) for a in ho.unpack(p0, 1024, 18, 3)]
    z = torch.tanh(xt @ W1 + b1) @ W2 + b2
    F.cross_entropy(z, torch.from_numpy(y)).backward()
    assert _rel(dx, xt.grad.numpy()) < 1e-4
    hd.adam_step(lr=1e-3)
    hd.reset_optimizer()
    assert hd.step_t == 0 and not hd.grad_view().any()


def test_transfer_learn_with_backprop_into_embedding(tmp_path):
    """The reference's two-phase call: frozen-embedding fit, then everything un-frozen with Adam(embedding_lr)."""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    data = make_fewshot_dataset(str(tmp_path / "d"))
    ms = input_data.standard_microspeech_model_settings(3)
    name, model, details = tl.transfer_learn(
        target="target", train_files=data["train"], val_files=data["val"], unknown_files=data["unknown"],
        num_epochs=2, num_batches=1, batch_size=16, primary_lr=0.001, backprop_into_embedding=True, embedding_lr=0.0001,
        model_settings=ms, base_model_path="synthetic", base_model_output="dense_2", UNKNOWN_PERCENTAGE=50.0,
        bg_datadir=data["bg_dir"], csvlog_dest=None, verbose=0, seed=5)
    assert set(details) == {"num_epochs", "batch_size", "num_batches", "val_accuracy", "target"}
    assert name == f"xfer_epochs_2_bs_16_nbs_1_val_acc_{details['val_accuracy']:0.2f}_target_target"
    h = model.history
    assert all(len(h[k]) == 2 and np.isfinite(h[k]).all() for k in ("loss", "accuracy", "val_loss", "val_accuracy"))
    # the embedding really moved: kernels by Adam, BatchNorm moving statistics by the training-mode forwards
    base = weights.synthetic_blob()
    tuned = model._blob
    T = {t["name"]: t for t in weights.manifest()}
    for nme in ("stem_conv/kernel", "block3a_dwconv/depthwise_kernel", "block6b_se_reduce/bias", "top_bn/gamma", "dense_2/kernel", "block2a_bn/moving_mean"):
        t = T[nme]
        assert not np.array_equal(tuned[t["offset"]:t["offset"] + t["count"]], base[t["offset"]:t["offset"] + t["count"]]), nme
    assert np.isfinite(tuned).all()
    # returned model: inference kernels on the fine-tuned weights == the inference oracle on the same weights
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    specs = np.stack([input_data.file2spec(ms, f) for f in data["val"]])
    preds = model.predict(specs[..., None])
    ref_probs, _ = ho.forward(model.head.get_params(), EmbeddingOracle(tuned).forward(specs).numpy())
    assert np.abs(preds - ref_probs).max() < 1e-3 and np.array_equal(preds.argmax(1), ref_probs.argmax(1))
    model.save(str(tmp_path / "m"))
    again = tl.TransferLearnedModel.load(str(tmp_path / "m"), max_batch=64)
    assert np.array_equal(again.predict(specs[..., None]), preds)


def test_bucketed_gradient_allreduce_over_rccl():
    """backward(allreduce=True) in a 1-rank "nccl" (= RCCL) group: three contiguous ranges of the flat gradient buffer
    (dense_1.., top_conv..dense_1, 0..top_conv) are all-reduced as they become final and cover the buffer exactly once."""
    import socket
    import torch.distributed as dist
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(1)
        spec = torch.from_numpy(rng.integers(0, 670, size=(4, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda()
        d_emb = torch.from_numpy(rng.standard_normal((4, 1024)).astype(np.float32)).cuda()
        tr = EmbeddingTrainer(weights.synthetic_blob())
        tr.forward_train(spec)
        tr.backward(d_emb)
        plain = tr.grads.clone()
        spans, real = [], dist.all_reduce
        tr2 = EmbeddingTrainer(weights.synthetic_blob())
        tr2.forward_train(spec)
        real_ptr = tr2.grads.data_ptr()

        def counting2(t, *a, **k):
            lo = (t.data_ptr() - real_ptr) // 4
            spans.append((lo, lo + t.numel()))
            return real(t, *a, **k)
        dist.all_reduce = counting2
        tr2.backward(d_emb, allreduce=True)
        dist.all_reduce = real
        assert len(spans) == 3
        spans.sort()
        assert spans[0][0] == 0 and spans[-1][1] == tr2.grads.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        # every reduction has a fixed order (round 3): a second trainer on the same inputs gives the same gradient bit for bit
        assert torch.equal(tr2.grads, plain)
    finally:
        dist.destroy_process_group()


def test_allreduce_ranges_are_final_when_the_collective_reads_them():
    """The weight gradients run on a second stream; a range's all-reduce must not start before they have landed (flush of the side
    context, main stream waits on the side stream, flush of the main context, THEN the collective).  A 1-rank SUM cannot show a missing
    dependency (it is a no-op), so the collective is replaced by a snapshot of the range taken on torch's current stream -- the stream
    RCCL orders itself behind -- at the moment all_reduce is called: every snapshot must equal the final gradient of a single-stream
    run, bit for bit (fixed-order reductions)."""
    import socket
    import torch.distributed as dist
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    real = dist.all_reduce
    try:
        rng = np.random.default_rng(2)
        B = 16
        spec = torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda()
        d_emb = torch.from_numpy(rng.standard_normal((B, 1024)).astype(np.float32)).cuda()
        ref = EmbeddingTrainer(weights.synthetic_blob())
        ref.overlap_wgrad = False
        ref.forward_train(spec)
        ref.backward(d_emb)
        torch.cuda.synchronize()
        want = ref.grads.clone()
        tr = EmbeddingTrainer(weights.synthetic_blob())
        assert tr.overlap_wgrad
        for rep in range(3):                     # the first pass also warms the side stream's context
            tr.forward_train(spec)
            base, snaps = tr.grads.data_ptr(), []

            class Done:
                def wait(self):
                    pass

            def snapshot(t, *a, **k):
                lo = (t.data_ptr() - base) // 4
                snaps.append((lo, t.clone()))
                return Done()
            dist.all_reduce = snapshot
            tr.backward(d_emb, allreduce=True)
            dist.all_reduce = real
            torch.cuda.synchronize()
            assert len(snaps) == 3 and sum(c.numel() for _, c in snaps) == want.numel()
            for lo, c in snaps:
                assert torch.equal(c, want[lo:lo + c.numel()]), (rep, lo)
            assert torch.equal(tr.grads, want)
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()


def test_graph_replayed_training_step_equals_the_eager_step():
    """TrainStepGraph: forward + head loss + backward + both Adam updates of one `backprop_into_embedding` step recorded once and
    replayed (the call tape by default, one hipGraph on request).  (a) Three replays with fresh inputs and drop-connect masks leave
    EXACTLY what the same launches issued one by one leave (use_graph=False: same kernels, same buffers, device-side step counter;
    one stream or two) -- every reduction has a fixed order.  (b) Against
    the host-driven eager step (EmbeddingTrainer / Head methods with host step counters) after ONE step: host pow vs device pow in lr_t
    differ in the last bit, and Adam turns a last-bit difference of a noise-level gradient (betas in front of a batch-statistics BN
    have an exactly-zero true gradient) into a full +-lr step, so a handful of parameters may sit lr apart; everything else agrees to
    round-off.  (Comparing the two after several steps is not a test of the code: a ReLU of the dense stack that flips for one clip
    under a 1e-7 perturbation changes a whole weight column's gradient -- seen in round 4 when a kernel's summation order changed.)"""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer, TrainStepGraph, drop_connect_rates
    from multilingual_kws_amd.head import Head
    blob = weights.synthetic_blob()
    rng = np.random.default_rng(11)
    B, lr = 8, 1e-4
    specs = [torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda() for _ in range(3)]
    labels = [torch.from_numpy(rng.integers(0, 3, B).astype(np.int32)).cuda() for _ in range(3)]
    masks = [{n: rng.uniform(0, 1, B) >= r for n, r in drop_connect_rates().items()} for _ in range(3)]
    # (a) graph replay == the same launches one by one, bit for bit, over three steps
    tr_l, hd_l = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    step_l = TrainStepGraph(tr_l, hd_l, B, lr, use_graph=False)
    stats_l = [step_l.run(x, y, mk).clone() for x, y, mk in zip(specs, labels, masks)]
    tr_g, hd_g = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    step = TrainStepGraph(tr_g, hd_g, B, lr)
    assert step._tape and step.graph is None and step_l.graph is None and step_l._tape is None        # default: the recorded call tape
    assert np.array_equal(tr_g.blob(), blob) and int(tr_g.d_step.item()) == 0           # the capture warm-up left no trace
    for i, (x, y, mk) in enumerate(zip(specs, labels, masks)):
        assert torch.equal(step.run(x, y, mk), stats_l[i]), i
    torch.cuda.synchronize()
    assert int(tr_g.d_step.item()) == 3 == int(tr_l.d_step.item())
    pg = tr_g.blob()
    assert np.array_equal(pg, tr_l.blob()) and np.array_equal(hd_g.get_params(), hd_l.get_params())
    # (b) one step of the host-driven eager path
    tr_e, hd_e = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    emb = tr_e.forward_train(specs[0], masks[0])
    st_e = hd_e.loss_grad(emb, labels[0]).clone()
    tr_e.backward(hd_e.input_grad(B))
    hd_e.adam_step(lr=lr)
    tr_e.adam_step(lr=lr)
    tr_1, hd_1 = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    st_1 = TrainStepGraph(tr_1, hd_1, B, lr).run(specs[0], labels[0], masks[0]).clone()
    assert torch.allclose(st_1, st_e, rtol=1e-4, atol=1e-4)
    diff = np.abs(tr_e.blob() - tr_1.blob())
    assert diff.max() <= 1.5 * lr and (diff > 1e-6).mean() < 1e-3
    assert np.abs(hd_e.get_params() - hd_1.get_params()).max() <= 1.5 * lr
    # and the graph path repeats itself bit for bit
    tr_h, hd_h = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    step2 = TrainStepGraph(tr_h, hd_h, B, lr)
    for x, y, mk in zip(specs, labels, masks):
        step2.run(x, y, mk)
    assert np.array_equal(tr_h.blob(), pg) and np.array_equal(hd_h.get_params(), hd_g.get_params())
    # the default is the recorded call tape (two streams); the single-stream hipGraph form of the step walks the same trajectory bit for bit
    tr_c, hd_c = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    step3 = TrainStepGraph(tr_c, hd_c, B, lr, use_graph=True)
    assert step2.mode == "tape" and step3.mode == "hipgraph" and step_l.mode == "eager"
    for x, y, mk in zip(specs, labels, masks):
        step3.run(x, y, mk)
    assert np.array_equal(tr_c.blob(), pg) and np.array_equal(hd_c.get_params(), hd_g.get_params())
    # and so does a trainer that keeps its weight gradients on the caller's stream
    tr_s, hd_s = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
    tr_s.overlap_wgrad = False
    step4 = TrainStepGraph(tr_s, hd_s, B, lr, use_graph=False)
    for x, y, mk in zip(specs, labels, masks):
        step4.run(x, y, mk)
    assert np.array_equal(tr_s.blob(), pg) and np.array_equal(hd_s.get_params(), hd_g.get_params())


def test_recorded_step_survives_another_batch_size_in_between():
    """The call tape names the trainer's pooled buffers.  A forward pass at ANOTHER batch size rebuilds that pool; the next replay must notice
    (re-record on the new buffers) instead of writing through stale pointers: same parameters as the same sequence run launch by launch."""
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer, TrainStepGraph
    from multilingual_kws_amd.head import Head
    blob = weights.synthetic_blob()
    rng = np.random.default_rng(21)
    B, lr = 8, 1e-4
    specs = [torch.from_numpy(rng.integers(0, 670, size=(B, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda() for _ in range(2)]
    labels = [torch.from_numpy(rng.integers(0, 3, B).astype(np.int32)).cuda() for _ in range(2)]
    other = torch.from_numpy(rng.integers(0, 670, size=(5, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda()
    out = []
    for mode in ("eager", "tape"):
        tr, hd = EmbeddingTrainer(blob), Head(max_batch=B, seed=3)
        step = TrainStepGraph(tr, hd, B, lr, mode=mode)
        step.run(specs[0], labels[0])
        tr.forward_train(other)                    # training-mode forward at batch 5: moving statistics move, the buffer pool is rebuilt
        tr.tape = None
        step.run(specs[1], labels[1])
        torch.cuda.synchronize()
        out.append((tr.blob(), hd.get_params()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_two_trainers_interleaved_on_one_thread_do_not_share_operator_state():
    """mkws_train_ctx (include/mkws.h): each EmbeddingTrainer owns its scratch arena and deferred-fold queue and binds them at the top of
    every public method.  Two trainers whose forward / backward calls are INTERLEAVED on one host thread -- the second one's forward
    runs while the first has a tape pending, its backward queues folds between the first's forward and backward -- must each produce
    exactly the gradient they produce alone; so must a trainer whose backward runs on another host thread than its forward."""
    import threading
    from multilingual_kws_amd import weights
    from multilingual_kws_amd.embedding_trainer import EmbeddingTrainer
    blob = weights.synthetic_blob()
    rng = np.random.default_rng(5)
    xs = [torch.from_numpy(rng.integers(0, 670, size=(4, 49, 40)).astype(np.float32) * np.float32(10 / 256)).cuda() for _ in range(2)]
    ds = [torch.from_numpy(rng.standard_normal((4, 1024)).astype(np.float32)).cuda() for _ in range(2)]

    def alone(k):
        tr = EmbeddingTrainer(blob)
        tr.forward_train(xs[k], None)
        tr.backward(ds[k])
        return tr.grads.clone()
    want = [alone(0), alone(1)]
    a, b = EmbeddingTrainer(blob), EmbeddingTrainer(blob)
    a.forward_train(xs[0], None)
    b.forward_train(xs[1], None)
    b.backward(ds[1])
    a.backward(ds[0])
    assert torch.equal(a.grads, want[0]) and torch.equal(b.grads, want[1])
    c = EmbeddingTrainer(blob)
    c.forward_train(xs[0], None)
    err = []

    def other_thread():
        try:
            with torch.cuda.device(0):
                c.backward(ds[0])
                torch.cuda.synchronize()
        except Exception as e:      # pragma: no cover
            err.append(e)
    t = threading.Thread(target=other_thread)
    t.start(); t.join()
    assert not err and torch.equal(c.grads, want[0])
