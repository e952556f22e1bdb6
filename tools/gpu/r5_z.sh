cd $GRAFT_REPO_ROOT
for v in "6 1" "8 1" "8 0" "4 1" "6 1" "6 0"; do
  set -- $v
  timeout 300 python bench.py --config finetune --no-cpu-baseline --ft-group $1 --ft-overlap $2 > /tmp/ft.json 2> /tmp/ft.err; echo "ft-group $1 overlap $2: $(python -c "
import json;d=json.load(open('/tmp/ft.json'));print(d['value'],d['ms_per_step'],d['whole_step'])" 2>&1 | tail -1)"
done
MKWS_FT_CPROFILE=0 timeout 300 python tools/finetune_group_profile.py 512 8 2>&1 | grep "B=512\|device"
