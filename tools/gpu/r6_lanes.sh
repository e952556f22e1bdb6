# headline config with 1 / 2 / 3 / 4 independent batches in flight (bench.py --lanes), same call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_lanes; rm -rf $O; mkdir -p $O
for l in 1 2 3 4 2 1; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --lanes $l > $O/embed_l$l.json 2> $O/embed_l$l.err
  python - "$O/embed_l$l.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('lanes',d.get('lanes'),'value',d['value'],'ms',d['ms_per_step'],'whole',d['roofline']['whole_step_frac'],'sustained',d.get('sustained'),'single',d.get('single_stream'))
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err; echo "rc=$?"; python -c "
import json;d=json.load(open('$O/driver_line.json'));print(d['value'],d['ms_per_step'],d.get('sustained'),d.get('single_stream'));print(json.dumps(d.get('secondary'),indent=1)[:3000]);print(d['cpu_baseline'])"
