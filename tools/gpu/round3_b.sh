cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_hf_efficientnet_golden.py tests/test_streaming.py tests/test_head_gpu.py tests/test_pipeline_gpu.py tests/test_consumers.py -m gpu -q -k "pair_exchange or serving or graph or hf or port or streaming or head or cut or transfer or consumer or multi" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 400 python bench.py --config stream --steps 30 --warmup 5 > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench rc=$?"; tail -3 $O/bench_stream.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3b/bench_stream.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("whole_step_frac"), d.get("latency_ms_batch1"), d.get("latency_ms_batch1_eager"))
print(d["whole_step"]); print(d["cpu_baseline"])
PY
