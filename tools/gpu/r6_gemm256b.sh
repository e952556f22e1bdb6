# dense tile rule for mid-size handles: parity of every handle size + tables + stream bench (old rule forced through MKWS_GEMM_FORCE)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gemm256b; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_embedding_gpu.py tests/test_streaming.py tests/test_surface.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for B in 256 512; do timeout 120 python tools/kernel_table.py $B 20 dense > $O/t$B.txt 2>&1; echo "B=$B: $(awk '/pass 1/{f=1} f && /dense/{s+=$(NF-1); printf "%s=%s ", $1, $(NF-1)} END{printf " sum=%.1f", s}' $O/t$B.txt)  $(grep 'pass 1' $O/t$B.txt)"; done
for rep in 1 2; do
  timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline > $O/stream_new.json 2> $O/stream_new.err
  MKWS_GEMM_FORCE=256,1,2,2 MKWS_GEMM_FORCE_MIN=256 timeout 300 python bench.py --config stream --steps 30 --warmup 4 --no-cpu-baseline > $O/stream_old.json 2> $O/stream_old.err
  python -c "
import json
for n in ('new','old'):
    d=json.load(open('$O/stream_%s.json'%n));print(n,d['value'],d['ms_per_step'],d['roofline']['whole_step_frac'],d.get('latency_ms_batch1'))"
done
