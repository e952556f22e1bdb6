# fused head training kernel (rows + dW1 partial sums in one launch) vs the two launches: tests + A/B by MKWS_HEAD_FUSED (bit-identical gradients)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_u; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_head_gpu.py tests/test_finetune_gpu.py tests/test_pipeline_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
MKWS_HEAD_FUSED=0 timeout 600 python -m pytest -m gpu -q tests/test_head_gpu.py > $O/pytest_unfused.log 2>&1; echo "pytest (unfused) rc=$?"; tail -1 $O/pytest_unfused.log
python - <<'PY'
# same gradient bits from both paths
import os, subprocess, sys
code = """
import numpy as np, torch, sys
sys.path.insert(0, '.')
from multilingual_kws_amd.head import Head
rng = np.random.default_rng(0)
for B in (512, 300, 64, 7):
    x = torch.from_numpy((rng.standard_normal((B, 1024)) * 0.3).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, 3, B).astype(np.int32)).cuda()
    h = Head(max_batch=512, seed=1)
    st = h.loss_grad(x, y).tolist()
    g = h.grad_view(with_stats=True).cpu().numpy()
    print(B, st, float(np.abs(g).sum()), hash(g.tobytes()))
"""
outs = []
for v in ("1", "0"):
    env = dict(os.environ, MKWS_HEAD_FUSED=v, PYTHONHASHSEED="0")
    outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout)
print(outs[0])
print("fused == two launches, bit for bit:", outs[0] == outs[1] and len(outs[0]) > 0)
PY
for rep in 1 2; do
for v in 1 0; do
  echo "== MKWS_HEAD_FUSED=$v"; MKWS_HEAD_FUSED=$v MKWS_FT_CPROFILE=0 timeout 300 python tools/finetune_group_profile.py 512 4 2>&1 | grep "B=512\|device"
done
done
