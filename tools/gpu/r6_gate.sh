cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gate; rm -rf $O; mkdir -p $O
python - <<'PY' 2>&1 | grep -v Warn | tee $O/equal.txt
import torch, numpy as np
from multilingual_kws_amd import weights
from multilingual_kws_amd.embedding_model import EmbeddingModel
blob = weights.synthetic_blob()
for mb in (1024, 512, 100):
    em = EmbeddingModel(blob, max_batch=mb)
    x = torch.rand((mb, 49, 40), device="cuda") * 26
    em.set_option("fuse_gate", 0); a = em.forward(x).clone(); ta = em.tap(x[:9], "block6c_gate").clone(); tb = em.tap(x[:9], "block7a").clone()
    em.set_option("fuse_gate", 1); b = em.forward(x).clone(); ua = em.tap(x[:9], "block6c_gate").clone(); ub = em.tap(x[:9], "block7a").clone()
    print(mb, "forward equal", torch.equal(a, b), "gate tap equal", torch.equal(ta, ua), "block tap equal", torch.equal(tb, ub), "finite", bool(torch.isfinite(b).all()))
PY
AB_OPTION=fuse_gate timeout 600 python tools/kernel_table.py 1024 20 chain 2>&1 | grep -E "chain|forward" | tee $O/ab1024.txt
AB_OPTION=fuse_gate timeout 600 python tools/kernel_table.py 256 20 chain 2>&1 | grep -E "pair_chain|forward" | tee $O/ab256.txt
