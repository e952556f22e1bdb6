cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "1 2" "0 0"; do
  set -- $v
  O=$GRAFT_REPO_ROOT/gpurun_out/r5_n_g$1; rm -rf $O; mkdir -p $O
  ( cd /tmp && MKWS_TRAIN_GEMM2=$1 MKWS_TRAIN_GEMM_TN2=$2 MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py 512 > $O/log.txt 2>&1 )
  echo "== gemm2=$1 tn2=$2"; grep "B=" $O/log.txt; f=$(find $O -name "*kernel_stats.csv" | head -1); python - $f <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', round(tot/1e6,2))
for r in rows[:14]:
    print('  %-70s calls %5s avg %7.1f us tot %6.2f ms'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
