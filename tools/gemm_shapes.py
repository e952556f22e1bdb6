"""Times mkws_op_gemm on the 1x1-conv / dense shapes of a training step (batch B): forward NN, input-gradient NT, weight-gradient TN.
   python tools/gemm_shapes.py [B]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import _lib
from multilingual_kws_amd.arch import BLOCKS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = _lib.lib()
dev = torch.device("cuda:0")
scratch = torch.empty(32 << 20, dtype=torch.float32, device=dev)
_lib.check(L.mkws_op_set_scratch(ctypes.c_void_p(scratch.data_ptr()), scratch.numel()))
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = _lib.current_stream_ptr()
shapes = []
H, W = 25, 20
for name, cin, cout, k, st, e in BLOCKS:
    ce = cin * e
    if e != 1:
        shapes.append((f"{name} expand", B * H * W, cin, ce))
    if st == 2:
        H, W = (H + 1) // 2, (W + 1) // 2
    shapes.append((f"{name} project", B * H * W, ce, cout))
shapes += [("top", B * H * W, 320, 1280), ("dense", B, 1280, 2048), ("dense_1", B, 2048, 2048), ("dense_2", B, 2048, 1024)]
tot = {"NN": 0.0, "NT": 0.0, "TN": 0.0}
print(f"B={B}:  layer  M K N | NN us TF/s | NT us TF/s | TN us TF/s")
for name, M, K, N in shapes:
    X, Wt, Z = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    dZ, dX, dW = torch.randn(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(K, N, device=dev)
    calls = {"NN": lambda: L.mkws_op_gemm(p(X), p(Wt), p(Z), M, N, K, K, N, N, 0, 0, 0, 0, s),
             "NT": lambda: L.mkws_op_gemm(p(dZ), p(Wt), p(dX), M, K, N, N, N, K, 0, 1, 0, 0, s),
             "TN": lambda: L.mkws_op_gemm(p(X), p(dZ), p(dW), K, N, M, K, N, N, 1, 0, 0, 0, s)}
    row = f"{name:12s} {M:7d} {K:5d} {N:5d} |"
    for kind, fn in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tot[kind] += us
        row += f" {us:7.1f} {2.0 * M * K * N / us / 1e6:6.1f} |"
    print(row)
print("sum us:", {k: round(v, 1) for k, v in tot.items()})
