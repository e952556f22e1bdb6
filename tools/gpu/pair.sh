# paired whole-block kernel: parity subset, same-call A/B bench, per-phase timing   gpurun -- 'bash tools/gpu/pair.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/pair; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py -x -q -k "options or full_batch or ragged" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for v in 1 0; do
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --opt fuse_pair=$v > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json
for v in (1, 0):
    d=json.load(open(f"gpurun_out/pair/bench_{v}.json"))
    print(v, d["value"], d["ms_per_step"], {k: x["ms_per_step"] for k, x in d["kernels"].items() if "pair" in k or "2,2,1,8" in k})
PY
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py 2>&1 | grep -E "block-timing\] block(6c|6d|7a)" | tail -3
