cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_serve2; rm -rf $O; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
MODES=graphs,eager,join LANES=4 BIG=1,2 timeout 300 python tools/lane_modes.py 256 2>&1 | grep max_batch | tee -a $O/modes.txt
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q MKWS_SERVING_LANES=4 timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline > $O/s4_q$q.json 2> $O/s4_q$q.err
  python -c "
import json;d=json.load(open('$O/s4_q$q.json'));print('queues $q serving lanes 4:',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'])" | tee -a $O/lanes.txt
done
