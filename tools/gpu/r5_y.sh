# batch-1 latency: cluster launches with helper workgroups that pull the next launch's weights into the XCD's L2 (option cluster_prefetch), A/B in one process
cd $GRAFT_REPO_ROOT
timeout 300 python tools/latency_ab.py cluster_prefetch 1 3 2>&1 | grep "round"
timeout 300 python tools/latency_ab.py cluster_prefetch 4 2 2>&1 | grep "round"
timeout 600 python -m pytest -m gpu -q tests/test_embedding_gpu.py -k "serving_handle_plans or guard" 2>&1 | tail -3
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0, '.')
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
blob = weights.synthetic_blob()
for mb in (1, 3, 8, 32):
    x = torch.rand((mb, 49, 40), device='cuda') * 26
    a, b = EmbeddingModel(blob, max_batch=mb), EmbeddingModel(blob, max_batch=mb)
    b.set_option("cluster_prefetch", 1)
    ra = a.forward(x); rb = b.forward(x); rb2 = b.forward(x)
    torch.cuda.synchronize()
    print("max_batch", mb, "prefetch on == off bit for bit:", torch.equal(ra, rb) and torch.equal(rb, rb2), "errors", a.get_option("exchange_error"), b.get_option("exchange_error"), b.get_option("pair_degraded"))
PY
