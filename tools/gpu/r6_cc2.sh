cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_cc2; rm -rf $O; mkdir -p $O
MKWS_LIB=$PWD/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/one_fwd_small.py 1 2>&1 | grep -E "cluster-chain" | tail -10 | tee $O/chain_timing.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o lat -- python $GRAFT_REPO_ROOT/tools/latency_profile.py 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P' | tee $O/latency_stats.txt
import csv, glob
f = glob.glob("gpurun_out/r6_cc2/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = 0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    per = float(r["TotalDurationNs"]) / 300 / 1e3
    tot += per
    print(f"{per:8.1f} us/window  x {int(r['Calls'])/300:5.1f} avg {float(r['AverageNs'])/1e3:6.1f}  {r['Name'][:100]}")
print("sum of kernel time per window", tot)
P
rm -rf $O/prof
timeout 600 python bench.py --config stream 2>/dev/null | tail -1 > $O/bench_stream.json; python -c "
import json; d=json.load(open('$O/bench_stream.json')); print({k:d[k] for k in ('value','ms_per_step') if k in d}); print({k:v for k,v in d.items() if 'lat' in k}); print(d.get('config'))"
