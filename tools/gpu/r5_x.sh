cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout 300 python bench.py --config finetune --no-cpu-baseline > /tmp/ft.json 2>/dev/null; python -c "
import json;d=json.load(open('/tmp/ft.json'));print('run $i:', d['value'], d['ms_per_step'], d['steps_per_forward'], d['head_steps_on_second_stream'])"
done
timeout 300 python bench.py --config finetune --no-cpu-baseline --steps 20 --warmup 5 | python -c "
import json,sys;d=json.load(sys.stdin);print('20/5:', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --config finetune --no-cpu-baseline --steps 60 --warmup 12 | python -c "
import json,sys;d=json.load(sys.stdin);print('60/12:', d['value'], d['ms_per_step'])"
