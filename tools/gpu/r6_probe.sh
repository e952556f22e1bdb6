cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_probe; rm -rf $O; mkdir -p $O
timeout 120 python tools/stream_probe.py 2>&1 | grep -v Warn | tee $O/q8.txt
GPU_MAX_HW_QUEUES=4 timeout 120 python tools/stream_probe.py 2>&1 | grep -v Warn | tee $O/q4.txt
GPU_MAX_HW_QUEUES=2 timeout 120 python tools/stream_probe.py 2>&1 | grep -v Warn | tee $O/q2.txt
timeout 600 python -m pytest tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for l in 4 5 6 4; do
  MKWS_SERVING_LANES=$l timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline > $O/s$l.json 2> $O/s$l.err
  python -c "
import json;d=json.load(open('$O/s$l.json'));print('serving lanes $l:',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'))" | tee -a $O/lanes.txt
done
GPU_MAX_HW_QUEUES=4 MKWS_SERVING_LANES=4 timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('4 hw queues, serving lanes 4:',d['value'],d['ms_per_step'],d.get('serving_lanes'))" | tee -a $O/lanes.txt
