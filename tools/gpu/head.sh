# head kernels: parity (head, fine-tune step, streaming, transfer_learn) + fine-tune / streaming bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/head; mkdir -p $O
timeout 600 python -m pytest tests/test_head_gpu.py tests/test_finetune_gpu.py tests/test_streaming.py tests/test_pipeline_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "passed|failed|rc=" $O/pytest.log | tail -2
for c in finetune stream; do
  timeout 300 python bench.py --no-cpu-baseline --steps 50 --config $c 2> $O/$c.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d.get('latency_ms_batch1'))"
done
