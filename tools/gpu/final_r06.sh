# Round-6 evidence in ONE call on ONE box / build: full -m gpu suite, kernel stats + PMC passes + the four bench lines (final.sh),
# rocprof kernel stats of the fine-tune / streaming bench commands, the training step, the batch-1 serving chain, the canonical 64-clip fine-tune.
#   gpurun --timeout 2400 -- 'bash tools/gpu/final_r06.sh'   then copy gpurun_out/final/* and gpurun_out/r06_* to profiles/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_pytest_gpu.log
bash tools/gpu/final.sh r06
R=r06 bash tools/gpu/stats_cfg.sh > gpurun_out/r06_stats_cfg.log 2>&1
export TMPDIR=/tmp
for B in 64 512; do
  O=$GRAFT_REPO_ROOT/gpurun_out/train_stats_$B; rm -rf $O; mkdir -p $O
  ( cd /tmp && MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py $B > $O/log.txt 2>&1 )
  cp $(find $O -name "*kernel_stats.csv" | head -1) gpurun_out/r06_kernel_stats_train$B.csv
done
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" > gpurun_out/r06_train_bench.txt; cat gpurun_out/r06_train_bench.txt
bash tools/gpu/latency_stats.sh 1 > gpurun_out/r06_latency_stats.txt 2>&1; cp $(find gpurun_out/lat_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r06_kernel_stats_latency.csv; tail -2 gpurun_out/r06_latency_stats.txt
# the reference's canonical fine-tune shape (transfer_learning.py: 64 clips per step): 16 optimizer steps per forward pass vs one
for g in 16 1; do
  timeout 300 python bench.py --config finetune --batch 64 --ft-group $g --no-cpu-baseline > gpurun_out/r06_variant_finetune64_g$g.json 2> gpurun_out/r06_variant_finetune64_g$g.err
  python -c "
import json;d=json.load(open('gpurun_out/r06_variant_finetune64_g$g.json'));print('finetune batch 64, steps per forward $g:', d['value'], 'clips/s', d['ms_per_step'], 'ms per optimizer step')"
done
timeout 300 python bench.py --config finetune --no-cpu-baseline --ft-group 1 > gpurun_out/r06_variant_finetune_g1.json 2>/dev/null
timeout 300 python tools/finetune_group_profile.py > gpurun_out/r06_finetune_group_profile.txt 2>&1
python - <<'PY'
import json
for c in ("embed","frontend","finetune","stream"):
    d=json.load(open(f"gpurun_out/final/r06_bench_{c}.json")); r=d["roofline"]
    print(c, d["value"], d["unit"], d["ms_per_step"], r["kernel"], r["frac"], r.get("whole_step_frac"), r.get("time_weighted_frac"), r["traffic"], d.get("latency_ms_batch1"), d.get("latency_ms_batch1_eager"), d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread"]["value"])
d=json.load(open("gpurun_out/r06_variant_finetune_g1.json")); print("finetune, one forward per step:", d["value"], d["ms_per_step"])
PY
