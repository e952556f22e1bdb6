# round 6, first call: the new parity tests (PNG pin, f2 on the device, element-wise tolerance) + a bench line of the round's starting point
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_first; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_tutorial_png_pin.py tests/test_checkpoint_device_gpu.py tests/test_embedding_gpu.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "embed rc=$? $(python -c "
import json;d=json.load(open('$O/embed.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac']);
for k,v in d['kernels'].items(): print('  %-50s %.4f ms  %.3f'%(k,v['ms_per_step'],v['frac']))")"
