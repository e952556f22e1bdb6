# whole -m gpu suite + A/B tables + all four bench lines (no rocprof)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_full; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for B in 1024 512 256; do
AB_OPTION=fuse_mid AB_VALUES=3,1 timeout 300 python tools/kernel_table.py $B 20 "" > $O/table_mid$B.txt 2>&1; grep -E "^pass" $O/table_mid$B.txt
done
bash tools/gpu/r4_chain_timing.sh 2>&1 | grep -v "block-timing"
for cfg in embed frontend finetune stream; do
timeout 600 python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline > $O/$cfg.json 2> $O/$cfg.err; echo "$cfg rc=$? $(python -c "
import json;d=json.load(open('$O/$cfg.json'));print(d['value'],d['ms_per_step'],d['roofline'].get('whole_step_frac'), d.get('latency_ms_batch1'))")"
done
