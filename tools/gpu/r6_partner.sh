cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_partner; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for p in 1 0; do
  echo "## MKWS_PARTNER_STREAM=$p" | tee -a $O/train.txt
  MKWS_PARTNER_STREAM=$p timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" | tee -a $O/train.txt
  MKWS_PARTNER_STREAM=$p timeout 300 python bench.py --config finetune --steps 240 --warmup 24 --no-cpu-baseline 2>$O/ft_err_$p.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('partner stream $p: finetune',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'])" | tee -a $O/ft.txt
done
done
timeout 900 python -m pytest tests/test_finetune_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
