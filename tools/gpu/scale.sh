# Weak-scaling curve on an 8-GPU MI355X node (the driver's SCALE run does the same; this is for whoever has such a node):
# BASELINE configs[2] (embedding forward, no data-path collective) and configs[3] (fine-tune step, ONE RCCL all-reduce of
# 18 509 floats per step) at N = 1, 2, 4, 8 ranks, efficiency(N) = value(N) / (N * value(1)).
#   bash tools/gpu/scale.sh [steps]      -> gpurun_out/scale/{embed,finetune}_N.json + scale.json
# bench.py started with --gpus N and no launcher re-launches itself under torch.distributed.run (one rank per GPU,
# rendezvous on 127.0.0.1) and refuses to report N GPUs from fewer devices -- it never fabricates a curve.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
STEPS=${1:-100}
O=gpurun_out/scale; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in embed finetune; do
  for n in 1 2 4 8; do
    timeout 900 python bench.py --config $c --gpus $n --steps $STEPS --warmup 20 --no-cpu-baseline > $O/${c}_$n.json 2> $O/${c}_$n.err
    echo "$c N=$n rc=$? $(head -c 200 $O/${c}_$n.json)"
  done
done
python - <<'PY'
import json
out = {}
for c in ("embed", "finetune"):
    vals = {}
    for n in (1, 2, 4, 8):
        try:
            vals[n] = json.load(open(f"gpurun_out/scale/{c}_{n}.json"))["value"]
        except Exception as e:
            vals[n] = None
    out[c] = {"value": vals, "efficiency": {n: (round(v / (n * vals[1]), 4) if v and vals.get(1) else None) for n, v in vals.items()}}
json.dump(out, open("gpurun_out/scale/scale.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
