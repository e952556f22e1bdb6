# round 6: the register-resident whole-block kernel (2b, 3b): parity first, then the per-kernel table with the option on / off in one process
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_rows; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py -x -q -m gpu -k "register_resident or every_stage or options_agree or full_batch" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
AB_OPTION=fuse_rows timeout 300 python tools/kernel_table.py 1024 20 block > $O/table.txt 2>&1; grep -E "pass|block2b|block3b" $O/table.txt
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "embed rc=$? $(python -c "
import json;d=json.load(open('$O/embed.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac']);
for k,v in d['kernels'].items(): print('  %-50s %.4f ms  %.3f'%(k,v['ms_per_step'],v['frac']))")"
