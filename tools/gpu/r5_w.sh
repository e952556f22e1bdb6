cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print('driver-style embed:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
