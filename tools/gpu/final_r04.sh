# Round-4 evidence in ONE call on ONE box / build: full -m gpu suite, kernel stats + PMC passes + the four bench lines (final.sh),
# rocprof kernel stats of the fine-tune / streaming bench commands and of the training step (tools/train_bench.py 64).
#   gpurun --timeout 1500 -- 'bash tools/gpu/final_r04.sh'   then copy gpurun_out/final/* and gpurun_out/r04_* to profiles/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_pytest_gpu.log
bash tools/gpu/final.sh r04
R=r04 bash tools/gpu/stats_cfg.sh > gpurun_out/r04_stats_cfg.log 2>&1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/train_stats; rm -rf $O; mkdir -p $O
( cd /tmp && MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py 64 > $O/log.txt 2>&1 )
cp $(find $O -name "*kernel_stats.csv" | head -1) gpurun_out/r04_kernel_stats_train64.csv
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" > gpurun_out/r04_train_bench.txt; cat gpurun_out/r04_train_bench.txt
# rocprof of the batch-1 serving chain (frontend -> embedding on the cluster plan -> 50 heads, 300 graph replays)
bash tools/gpu/latency_stats.sh 1 > gpurun_out/r04_latency_stats.txt 2>&1; cp $(find gpurun_out/lat_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r04_kernel_stats_latency.csv; tail -3 gpurun_out/r04_latency_stats.txt
python - <<'PY'
import json
for c in ("embed","frontend","finetune","stream"):
    d=json.load(open(f"gpurun_out/final/r04_bench_{c}.json")); r=d["roofline"]
    print(c, d["value"], d["unit"], d["ms_per_step"], r["kernel"], r["frac"], r.get("whole_step_frac"), r.get("time_weighted_frac"), r["traffic"], d.get("latency_ms_batch1"), d.get("latency_ms_batch1_eager"), d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread"]["value"])
PY
