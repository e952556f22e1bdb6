"""mkws_op_gemm time vs explicit ksplit (0 = the library's own choice) on a few training shapes.   python tools/gemm_ksplit.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev)
_lib.check(L.mkws_op_set_scratch(ctypes.c_void_p(scratch.data_ptr()), scratch.numel()))
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = _lib.current_stream_ptr()
SHAPES = [("4b proj B512", 6144, 480, 80), ("5a proj B512", 6144, 480, 112), ("3b proj B512", 17920, 240, 40), ("4a proj B512", 6144, 240, 80),
          ("4b exp B512", 6144, 80, 480), ("4b proj B64", 768, 480, 80), ("3b proj B64", 2240, 240, 40), ("4a proj B64", 768, 240, 80)]
if len(sys.argv) > 1 and sys.argv[1] == "dense":
    SHAPES = [("dense_1 B512", 512, 2048, 2048), ("dense B512", 512, 1280, 2048), ("dense_2 B512", 512, 2048, 1024), ("7a proj B512", 2048, 1152, 320),
              ("6b proj B512", 2048, 1152, 192), ("5b proj B512", 6144, 672, 112), ("6b exp B512", 2048, 192, 1152), ("top B512", 2048, 320, 1280),
              ("dense_1 B64", 64, 2048, 2048), ("6b proj B64", 256, 1152, 192), ("5b proj B64", 768, 672, 112), ("6b exp B64", 256, 192, 1152)]
for name, M, K, N in SHAPES:
    X, Wt, Z = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    dZ, dX = torch.randn(M, N, device=dev), torch.empty(M, K, device=dev)
    for kind in ("NN", "NT"):
        row = f"{name:14s} {kind} M{M} K{K if kind == 'NN' else N} N{N if kind == 'NN' else K} tiles {((M + 63) // 64) * (((N if kind == 'NN' else K) + 63) // 64):4d} |"
        for ks in (0, 1, 2, 3, 4, 6, 8):
            if kind == "NN":
                fn = lambda: L.mkws_op_gemm(p(X), p(Wt), p(Z), M, N, K, K, N, N, 0, 0, 0, ks, s)
            else:
                fn = lambda: L.mkws_op_gemm(p(dZ), p(Wt), p(dX), M, K, N, N, N, K, 0, 1, 0, ks, s)
            for _ in range(3):
                _lib.check(fn())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            row += f" ks{ks}:{e0.elapsed_time(e1) * 100:6.1f}"
        print(row)
