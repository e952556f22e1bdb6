# early-block kernel work: parity taps, per-CU workgroup timelines (timing build), bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/early; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_hf_efficientnet_golden.py tests/test_pipeline_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-220
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py > $O/out.log 2> $O/err.log; echo "rc=$?"
N=$(grep -n "wg-trace\] block1a" $O/err.log | tail -1 | cut -d: -f1)
tail -n +$N $O/err.log | grep "wg-trace\|front-timing\|mid-timing" | cut -c1-420
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "embed rc=$? $(python -c "
import json;d=json.load(open('$O/embed.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac']);
for k,v in d['kernels'].items(): print('  %-50s %.4f ms  %.3f'%(k,v['ms_per_step'],v['frac']))")"
