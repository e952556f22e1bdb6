# chain after the SGPR fix: parity subset, phase profile, A/B table, bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_chain2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py -m gpu -q -x -k "chain or options or ragged or recovers or graph_replay or exchange" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
bash tools/gpu/r4_chain_timing.sh
AB_OPTION=fuse_chain timeout 300 python tools/kernel_table.py 1024 20 "" > $O/table1024.txt 2>&1; grep -E "pass|chain|block_kernel" $O/table1024.txt
AB_OPTION=fuse_chain timeout 300 python tools/kernel_table.py 512 20 "" > $O/table512.txt 2>&1; grep -E "pass|chain" $O/table512.txt
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "bench rc=$?"; cat $O/embed.json | python -c "
import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'])
for k,v in list(d['kernels'].items())[:6]: print('  %-50s %.4f ms  %.3f'%(k,v['ms_per_step'],v['frac']))"
AB_OPTION=fuse_mid AB_VALUES=1,2 timeout 300 python tools/kernel_table.py 1024 20 "" > $O/table_mid.txt 2>&1; grep -E "pass|block2a|block2b|block3b" $O/table_mid.txt
