cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_queues2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_streaming.py tests/test_surface.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for q in default 3 2 default; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline 2>$O/err_$q.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('hw queues $q, serving lanes asked',d.get('serving_lanes'),'used',d.get('serving_lanes_used'),':',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'])" | tee -a $O/lanes.txt
done
