cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hf_efficientnet_golden.py tests/test_streaming.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for L in 1 2 4 6; do
  MKWS_SERVING_LANES=$L timeout 300 python bench.py --config stream --steps 30 --warmup 5 --no-cpu-baseline > $O/stream_l$L.json 2> $O/stream_l$L.err; echo "lanes $L rc=$? $(python -c "import json;d=json.load(open('$O/stream_l$L.json'));print(d['value'],d['ms_per_step'],d.get('latency_ms_batch1'))")"
done
MKWS_SERVING_LANES=4 timeout 300 python bench.py --config stream --steps 30 --warmup 5 --no-cpu-baseline --opt fuse_block=2 --opt fuse_mid=1 --opt fuse_back=1 > $O/stream_wb.json 2> $O/stream_wb.err; echo "whole-block lanes 4 rc=$? $(python -c "import json;d=json.load(open('$O/stream_wb.json'));print(d['value'],d['ms_per_step'])")"
MKWS_SERVING_LANES=1 timeout 300 python bench.py --config stream --steps 30 --warmup 5 --no-cpu-baseline --opt fuse_block=2 --opt fuse_mid=1 --opt fuse_back=1 > $O/stream_wb1.json 2> $O/stream_wb1.err; echo "whole-block lanes 1 rc=$? $(python -c "import json;d=json.load(open('$O/stream_wb1.json'));print(d['value'],d['ms_per_step'])")"
