# per-kernel stats of the training step at batch 64 (launch by launch; 13 steps + 11 forward passes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_train4; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/$O/stats; mkdir -p $S
( cd /tmp && MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $S -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py ${1:-64} > $S/log.txt 2>&1 )
cp $(find $S -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train.csv
grep "B=" $S/log.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4_train4/kernel_stats_train.csv')))
print('launches', sum(int(r['Calls']) for r in rows), 'total ms', sum(float(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:22]: print('%-60s %6s %8.1f us  %5.1f%%'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
