cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests/test_embedding_gpu.py -x -q > gpurun_out/r2h/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2h/pytest.log
tail -12 gpurun_out/r2h/pytest.log
for v in 1 0; do
timeout 300 python bench.py --no-cpu-baseline --steps 30 --opt split_block=$v > gpurun_out/r2h/bench_sb$v.json 2> gpurun_out/r2h/bench_sb$v.err
done
python - <<'PY'
import json
for n in ("sb1","sb0"):
    d=json.load(open(f"gpurun_out/r2h/bench_{n}.json"))
    print(n, d["value"], d["ms_per_step"])
    for k,v in d["kernels"].items():
        if any(t in k for t in ("xdw","gproj","splitk","se_expand<3","se_reduce<3","block_kernel<5,1,2","block_kernel<3,1,2")): print("   ", k, v)
PY
