cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py -m gpu -q -x -k "cluster or serving or session" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-220
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 200 python tools/one_fwd_small.py 1 2>&1 | grep cluster-timing | tail -11
bash tools/gpu/latency_stats.sh 1 2>&1 | head -9
timeout 300 python bench.py --config stream --steps 20 --warmup 5 --no-cpu-baseline > $O/stream.json 2> $O/stream.err; echo "stream rc=$? $(python -c "import json;d=json.load(open('$O/stream.json'));print(d['value'],d['ms_per_step'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'))")"
