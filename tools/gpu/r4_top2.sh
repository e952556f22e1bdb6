cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_top2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py -m gpu -q -x -k "chain or failed_pair or options or ragged or recovers or serving_handle or graph or every_stage or full_batch" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-200
for B in 1024 512; do
AB_OPTION=fuse_top AB_VALUES=0,1 timeout 300 python tools/kernel_table.py $B 20 "" > $O/table$B.txt 2>&1; grep -E "pass|pair_chain|top " $O/table$B.txt
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "bench rc=$?"; cat $O/embed.json | python -c "
import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'])
for k,v in list(d['kernels'].items())[:5]: print('  %-50s %.4f ms  %.3f'%(k,v['ms_per_step'],v['frac']))"
