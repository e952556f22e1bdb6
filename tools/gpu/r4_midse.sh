# mid kernel: SE weights requested in front of the chunk loop (default) vs at the top of the SE phase (timing build, MKWS_ABLATE=16)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_midse; rm -rf $O; mkdir -p $O
for A in 16 0 16 0; do
MKWS_ABLATE=$A MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd.py > $O/out$A.log 2> $O/err$A.log; echo "ablate=$A rc=$?"
N=$(grep -n "wg-trace\] block1a" $O/err$A.log | tail -1 | cut -d: -f1)
tail -n +$N $O/err$A.log | grep "mid-timing\|wg-phase.*block2b\|wg-phase.*block3a\|wg-phase.*block4a" | cut -c1-260
done
timeout 900 python -m pytest tests/test_embedding_gpu.py -m gpu -q -x -k "options or ragged or golden or oracle or taps or full_batch or every_stage" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);r=d['roofline'];print('embed', d['value'], d['ms_per_step'], r.get('whole_step_frac'))
for k,v in d['kernels'].items():
    if 'mid' in k: print('  ',k, round(v['ms_per_step']*1e3,1))"
