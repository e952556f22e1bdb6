"""One GEMM shape of the training step, repeated (for rocprofv3 --kernel-trace --stats):  python tools/gemm_one.py M K N kind[NN|NT|TN] [reps]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multilingual_kws_amd import _lib

M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
L = _lib.lib()
dev = torch.device("cuda:0")
scratch = torch.empty(32 << 20, dtype=torch.float32, device=dev)
_lib.check(L.mkws_op_set_scratch(ctypes.c_void_p(scratch.data_ptr()), scratch.numel()))
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = _lib.current_stream_ptr()
X, Wt, Z = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
dZ, dX, dW = torch.randn(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(K, N, device=dev)
calls = {"NN": lambda: L.mkws_op_gemm(p(X), p(Wt), p(Z), M, N, K, K, N, N, 0, 0, 0, 0, s),
         "NT": lambda: L.mkws_op_gemm(p(dZ), p(Wt), p(dX), M, K, N, N, N, K, 0, 1, 0, 0, s),
         "TN": lambda: L.mkws_op_gemm(p(X), p(dZ), p(dW), K, N, M, K, N, N, 1, 0, 0, 0, s)}
for _ in range(reps):
    _lib.check(calls[kind]())
torch.cuda.synchronize()
