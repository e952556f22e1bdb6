cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py tests/test_finetune_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-220
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 200 python tools/one_fwd_small.py 1 2>&1 | grep cluster-timing | tail -11
timeout 300 python bench.py --config stream --steps 20 --warmup 5 --no-cpu-baseline > $O/stream.json 2> $O/stream.err; echo "stream rc=$? $(python -c "import json;d=json.load(open('$O/stream.json'));print(d['value'],d['ms_per_step'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'))")"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "embed rc=$? $(python -c "import json;d=json.load(open('$O/embed.json'));print(d['value'],d['ms_per_step'])")"
