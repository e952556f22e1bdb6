"""Which of the first candidate streams overlap with each other?  Prints, for the runtime's current GPU_MAX_HW_QUEUES, the wall time of one
spin kernel per stream on growing sets of freshly created streams, and what multilingual_kws_amd.streams.concurrent_streams() picks.
    GPU_MAX_HW_QUEUES=4 python tools/stream_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multilingual_kws_amd  # noqa: F401  (sets the package's GPU_MAX_HW_QUEUES default)
import torch
from multilingual_kws_amd import streams as st

dev = torch.device("cuda:0")
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
st._spin_all(torch, cands[:1], dev)
one = min(st._spin_all(torch, cands[:1], dev) for _ in range(3))
print(f"one spin kernel: {one * 1e3:.3f} ms")
for k in range(2, 13):
    t = min(st._spin_all(torch, cands[:k], dev) for _ in range(2))
    print(f"first {k:2d} fresh streams together: {t * 1e3:.3f} ms = {t / one:.2f} x one")
pair = [[min(st._spin_all(torch, [cands[i], cands[j]], dev) for _ in range(2)) / one for j in range(8)] for i in range(8)]
print("pairwise (x one), streams 0..7:")
for i in range(8):
    print("  " + " ".join(f"{pair[i][j]:4.1f}" if i != j else "   -" for j in range(8)))
got = st.concurrent_streams(int(sys.argv[1]) if len(sys.argv) > 1 else 6, dev)
t = min(st._spin_all(torch, got, dev) for _ in range(2))
print(f"concurrent_streams -> {len(got)} streams, together {t * 1e3:.3f} ms = {t / one:.2f} x one")
