cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep -v amdgpu.ids | tail -6
export TMPDIR=/tmp
( cd /tmp && MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py 64 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1 )
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r03_kernel_stats_train64.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03_kernel_stats_train64.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("total ms", tot/1e6, "calls", calls)
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/1e3:10.0f} us {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%  {r['Name'][:90]}")
PY
