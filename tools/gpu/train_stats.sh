cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/train_stats; rm -rf $O; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py 64 > $O/log.txt 2>&1 )
tail -2 $O/log.txt
f=$(find $O -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/train_kernel_stats_64.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/train_kernel_stats_64.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e3:10.0f} us {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%  {r['Name'][:100]}")
PY
