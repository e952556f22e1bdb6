cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_slanes; rm -rf $O; mkdir -p $O
for l in 4 1 2 3 6 8 4; do
  MKWS_SERVING_LANES=$l timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline > $O/s$l.json 2> $O/s$l.err
  python -c "
import json;d=json.load(open('$O/s$l.json'));print('serving lanes $l:',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d.get('latency_ms_batch1'),d.get('latency_ms_batch1_eager'))"
done
