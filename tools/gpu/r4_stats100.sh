# rocprofv3 kernel stats of a LONGER bench run (100 + 10 steps): do the rocprof averages of the long kernels agree with the bench's hipEvent times then?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/stats100; rm -rf $O; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/err.log )
cp $(find $O -name "*kernel_stats.csv" | head -1) gpurun_out/r04_kernel_stats.csv
python - <<'PY'
import csv, json
rows=list(csv.DictReader(open('gpurun_out/r04_kernel_stats.csv')))
for r in rows[:6]: print('%-60s %5s %8.1f us'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
d=json.load(open('gpurun_out/stats100/bench.json'))
print('bench under rocprof:', d['value'], d['ms_per_step'], {k: round(v['ms_per_step']*1e3,1) for k,v in list(d['kernels'].items())[:3]})
PY
