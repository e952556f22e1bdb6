cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_serve; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_streaming.py tests/test_surface.py tests/test_consumers.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
MODES=graphs,eager LANES=4 BIG=1,2 timeout 300 python tools/lane_modes.py 256 2>&1 | grep max_batch | tee -a $O/modes.txt
for l in 4 1 2 3 4 5 6; do
  MKWS_SERVING_LANES=$l timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline > $O/s$l.json 2> $O/s$l.err
  python -c "
import json;d=json.load(open('$O/s$l.json'));print('serving lanes $l:',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'))" | tee -a $O/lanes.txt
done
