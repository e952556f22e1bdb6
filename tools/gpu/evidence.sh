# Round evidence on one MI355X: bench lines of all four BASELINE configs (with CPU baselines), rocprofv3 kernel stats of
# the default bench command, and three separate PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) over tools/one_fwd.py.
#   gpurun -- 'bash tools/gpu/evidence.sh'   then   python tools/pmc_to_json.py gpurun_out/evidence r02
cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence
rm -rf $O; mkdir -p $O
if [ -z "$SKIP_BENCH" ]; then
for c in embed frontend finetune stream; do
  timeout 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"
done
fi
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --sustain-s 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_write -o w -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1 )
for BS in 3072 512 256; do
( cd /tmp && ONE_FWD_B=$BS timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch_$BS -o f -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_fetch_$BS.log 2>&1 )
( cd /tmp && ONE_FWD_B=$BS timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_write_$BS -o w -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_write_$BS.log 2>&1 )
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -f csv -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o m -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $GRAFT_REPO_ROOT/$O/pmc_mfma.log 2>&1 )
find $O -name "*.csv" | head -20
python - <<'PY'
import csv, glob, json
for c in ("embed","frontend","finetune","stream"):
    try:
        d=json.load(open(f"gpurun_out/evidence/bench_{c}.json")); print(c, d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("latency_ms_batch1"), (d["cpu_baseline"] or {}).get("value"))
    except Exception as e: print(c, "failed", e)
f=glob.glob("gpurun_out/evidence/stats/**/*kernel_stats.csv", recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0]))); tot=0
    for r in rows[:14]:
        print(f"{float(r['TotalDurationNs'])/13/1000:8.1f} us/step avg {float(r['AverageNs'])/1000:7.1f}  {r['Name'][:80]}")
PY
