# Round-4 evidence refresh for the training operators only (mkws_train.hip changed after tools/gpu/final_r04.sh; the inference sources -- and
# with them the bench lines, kernel stats and PMC traffic of that call -- did not: bench.source_hash()): full -m gpu suite, smoke(), step times,
# rocprof kernel stats of 13 launch-by-launch steps + 11 forward passes at batch 64.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r04_pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python tools/train_bench.py 64 512 2>&1 | grep "B=" > gpurun_out/r04_train_bench.txt; cat gpurun_out/r04_train_bench.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/train_stats; rm -rf $O; mkdir -p $O
( cd /tmp && MKWS_TRAIN_BENCH_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O -o s -- python $GRAFT_REPO_ROOT/tools/train_bench.py 64 > $O/log.txt 2>&1 )
cp $(find $O -name "*kernel_stats.csv" | head -1) gpurun_out/r04_kernel_stats_train64.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04_kernel_stats_train64.csv')))
print('launches', sum(int(r['Calls']) for r in rows), 'kernel ms', round(sum(float(r['TotalDurationNs']) for r in rows)/1e6,1))
PY
timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);r=d['roofline'];print('embed', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['traffic'], r.get('whole_step_frac'), r.get('traffic_source'))"
