cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; rm -rf $O; mkdir -p $O
timeout 300 python tools/train_graph_probe.py 64 512 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py tests/test_finetune_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for c in stream finetune embed; do
timeout 300 python bench.py --config $c --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$? $(python -c "import json;d=json.load(open('$O/bench_$c.json'));print(d['value'],d['ms_per_step'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'),d['roofline']['kernel'],d['roofline']['frac'],d['roofline']['whole_step_frac'])")"
done
