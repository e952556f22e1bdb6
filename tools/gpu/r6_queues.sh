cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_queues; rm -rf $O; mkdir -p $O
GPU_MAX_HW_QUEUES=8 timeout 120 python tools/stream_probe.py 8 2>&1 | grep concurrent_streams | tee $O/probe.txt
for rep in 1 2; do
for q in 4 8 6 16 3; do
  GPU_MAX_HW_QUEUES=$q MKWS_SERVING_LANES=4 timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline 2>$O/err_$q.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('hw queues $q, serving lanes asked 4 used',d.get('serving_lanes_used'),':',d['value'],d['ms_per_step'])" | tee -a $O/lanes.txt
done
done
GPU_MAX_HW_QUEUES=4 MKWS_SERVING_LANES=8 timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline 2>$O/err_l8.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('hw queues 4, serving lanes asked 8 used',d.get('serving_lanes_used'),':',d['value'],d['ms_per_step'])" | tee -a $O/lanes.txt
