# frontend: bit-exact tests, then the frontend bench line at 4 waves per clip and at the library's choice, then the embed line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_fe; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_pipeline_gpu.py tests/test_consumers.py tests/test_streaming.py tests/test_finetune_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
for rep in 1 2 3; do for W in 4; do
MKWS_FRONTEND_WAVES=$W timeout 120 python bench.py --config frontend --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);r=d['roofline'];print('frontend waves=$W', d['value'], d['ms_per_step'], r['kernel'], r['frac'])"
done; done
timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);r=d['roofline'];print('embed', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('whole_step_frac'))"
timeout 120 python bench.py --config finetune --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);r=d['roofline'];print('finetune', d['value'], d['ms_per_step'])"
