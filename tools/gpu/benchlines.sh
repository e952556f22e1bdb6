# the four BASELINE bench lines only (run after tools/pmc_to_json.py so that roofline.traffic comes from the same build)
#   gpurun -- 'bash tools/gpu/benchlines.sh'   then copy gpurun_out/benchlines/bench_*.json to profiles/rNN_bench_*.json
cd $GRAFT_REPO_ROOT
O=gpurun_out/benchlines; rm -rf $O; mkdir -p $O
for c in embed frontend finetune stream; do
  timeout 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"
done
python - <<'PY'
import json
for c in ("embed","frontend","finetune","stream"):
    d=json.load(open(f"gpurun_out/benchlines/bench_{c}.json")); print(c, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
