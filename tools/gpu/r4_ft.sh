cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ft; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_finetune_gpu.py tests/test_pipeline_gpu.py tests/test_train_embedding_gpu.py tests/test_hf_efficientnet_train_golden.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/finetune_host_profile.py 2>&1 | cut -c1-200 | head -8
timeout 600 python bench.py --config finetune --steps 200 --warmup 20 --no-cpu-baseline > $O/finetune.json 2> $O/finetune.err; echo "finetune rc=$? $(python -c "
import json;d=json.load(open('$O/finetune.json'));print(d['value'],d['ms_per_step'],d['roofline'].get('whole_step_frac'), d['whole_step'])")"
