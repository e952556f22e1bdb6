# int16 hand-over augmentation -> frontend: parity + fine-tune bench A/B (float hand-over via MKWS_AUG_FLOAT=1), same call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_i16; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_finetune_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do for f in 1 0; do
  MKWS_AUG_FLOAT=$f timeout 300 python bench.py --config finetune --steps 240 --warmup 48 --no-cpu-baseline > $O/ft_f$f.json 2> $O/ft_f$f.err
  python -c "
import json;d=json.load(open('$O/ft_f$f.json'));print('float hand-over' if $f else 'int16 hand-over',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d['whole_step'])"
done; done
