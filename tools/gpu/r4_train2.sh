# training-step iteration: operator tests + whole-network gradient tests, step time, per-kernel stats of 13 steps at batch 64
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_train2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_hf_efficientnet_train_golden.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" > $O/train_bench.txt; cat $O/train_bench.txt
echo "-- weight gradients on the main stream"; MKWS_TRAIN_WGRAD_STREAM=0 timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" > $O/train_bench_nostream.txt; cat $O/train_bench_nostream.txt
export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/$O/stats; mkdir -p $S
( cd /tmp && MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $S -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py 64 > $S/log.txt 2>&1 )
cp $(find $S -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train64.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4_train2/kernel_stats_train64.csv')))
print('launches', sum(int(r['Calls']) for r in rows), 'total ms', sum(float(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:16]: print('%-60s %6s %8.1f us'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
