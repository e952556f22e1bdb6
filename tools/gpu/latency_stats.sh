cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "${@:-1}"; do
O=$GRAFT_REPO_ROOT/gpurun_out/lat_stats; rm -rf $O; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/latency_profile.py $cfg > $O/log.txt 2>&1 )
echo "== $cfg"
python - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/lat_stats/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=301.0
for r in rows[:26]:
    per=float(r['TotalDurationNs'])/n/1e3
    print(f"{per:7.1f} us/window  x{int(r['Calls'])/n:5.1f} avg {float(r['AverageNs'])/1e3:6.1f}  {r['Name'][:84]}")
print("sum of kernel time per window", sum(float(r['TotalDurationNs']) for r in rows)/n/1e3)
PY
done
