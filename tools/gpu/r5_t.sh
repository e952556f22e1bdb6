cd $GRAFT_REPO_ROOT
for g in 2 4 8 2 4; do
  timeout 300 python bench.py --config finetune --no-cpu-baseline --ft-group $g > /tmp/ft_$g.json 2> /tmp/ft_$g.err; echo "ft-group $g rc=$? $(python -c "
import json;d=json.load(open('/tmp/ft_$g.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d['whole_step'])" 2>&1 | tail -1)"
done
timeout 300 python bench.py --config embed --batch 2048 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print('embed batch 2048:', d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'])"
timeout 300 python bench.py --config stream --batch 1024 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print('stream batch 1024:', d['value'], d['ms_per_step'])"
