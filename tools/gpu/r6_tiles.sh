# 4x3-image kernels with one clip per workgroup on 256-clip handles: parity + per-kernel table by block_tiles + stream bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_tiles; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_embedding_gpu.py -x -q -m gpu -k "serving_handle_plans" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
AB_OPTION=block_tiles AB_VALUES=2,1 timeout 300 python tools/kernel_table.py 256 20 chain > $O/table.txt 2>&1; grep -E "pass|chain" $O/table.txt | tail -12
for t in 2 1; do
  timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline --opt block_tiles=$t > $O/stream_t$t.json 2> $O/stream_t$t.err
  python -c "
import json;d=json.load(open('$O/stream_t$t.json'));print('block_tiles $t:',d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d['roofline']['kernel'],d['roofline']['frac'],d.get('latency_ms_batch1'))"
done
