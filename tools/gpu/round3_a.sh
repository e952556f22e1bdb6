# round 3, call A: full -m gpu suite, default bench line, per-phase timing of the fused kernels (timing build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 400 python bench.py --steps 100 --warmup 20 > $O/bench_embed.json 2> $O/bench_embed.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a/bench_embed.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("whole_step_frac"), d["roofline"].get("time_weighted_frac"))
for k,v in d["kernels"].items(): print(f"{v['ms_per_step']*1000:8.1f} us  x{v['launches']}  frac {v['frac']:.3f}  {k}")
print(d["cpu_baseline"])
PY
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py 2> $O/timing.txt; grep -E "timing" $O/timing.txt | tail -40
