# round 5, call C: failing tests of call B again + host profile of the grouped fine-tune step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_surface.py tests/test_pipeline_gpu.py tests/test_finetune_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 300 python tools/finetune_group_profile.py > $O/ft_profile.txt 2>&1; head -40 $O/ft_profile.txt
for i in 1 2 3; do
timeout 300 python bench.py --config finetune --no-cpu-baseline > $O/ft_$i.json 2> $O/ft_$i.err
echo "finetune run $i rc=$? $(python -c "
import json;d=json.load(open('$O/ft_$i.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d['whole_step'])")"
done
