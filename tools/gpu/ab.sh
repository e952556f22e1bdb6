# same-call A/B of an execution option:  gpurun -- 'bash tools/gpu/ab.sh fuse_back 1 0'
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_$1; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
shift_opt=$1; shift
for v in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 30 --opt $shift_opt=$v > $O/bench_$v.json 2> $O/bench_$v.err
done
python - "$O" "$@" <<'PY'
import json, sys
O=sys.argv[1]
for v in sys.argv[2:]:
    d=json.load(open(f"{O}/bench_{v}.json"))
    print(v, d["value"], d["ms_per_step"])
    for k,x in d["kernels"].items():
        if any(t in k for t in ("back","true>","se_","pair","2,2,1,8")): print("   ", k, x)
PY
