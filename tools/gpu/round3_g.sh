cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_head_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python bench.py --config stream --steps 30 --warmup 5 --no-cpu-baseline > $O/stream.json 2> $O/stream.err; echo "stream rc=$? $(python -c "import json;d=json.load(open('$O/stream.json'));print(d['value'],d['ms_per_step'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'))")"
