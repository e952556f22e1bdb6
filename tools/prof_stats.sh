cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_c -o c -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_c.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_c/c_kernel_stats.csv')))
tot=0
for r in rows:
    per=float(r['TotalDurationNs'])/18/1000; tot+=per
    print(f"{per:8.1f} us/step {int(r['Calls'])//18:3d}x avg {float(r['AverageNs'])/1000:7.1f}  {r['Name'][:70]}")
print("total", tot)
PY
