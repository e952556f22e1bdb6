# verdict item 8: the paired chain's HBM write traffic -- exchange stores with the non-temporal hint vs plain (library A/B): timing + WRITE_SIZE / FETCH_SIZE passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
KFILTER=chain bash tools/gpu/ablibs.sh - multilingual_kws_amd/lib/libmkws_hip_ntstores.so - multilingual_kws_amd/lib/libmkws_hip_ntstores.so
for l in shipped ntstores; do
  if [ $l = shipped ]; then unset MKWS_LIB; else export MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_ntstores.so; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    O=$GRAFT_REPO_ROOT/gpurun_out/r5_r_${l}_$c; rm -rf $O; mkdir -p $O
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d $O -o p -- python $GRAFT_REPO_ROOT/tools/one_fwd.py > $O/log.txt 2>&1 )
    f=$(find $O -name "*counter_collection.csv" | head -1)
    python - $f $c $l <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if "pair_chain" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"][:60]][0] += float(r["Counter_Value"]); acc[r["Kernel_Name"][:60]][1] += 1
for k, (v, n) in acc.items():
    # counter units: WRITE_SIZE in 32 B; FETCH_SIZE in 32 B (x2 on gfx950 for wide reads per the guide is applied by tools/pmc_to_json.py; raw here)
    print(f"{sys.argv[3]:9s} {sys.argv[2]:10s} {k}: raw counter per launch {v / max(n,1):.0f}  (x32 B = {v / max(n,1) * 32 / 1e6:.1f} MB)")
PY
  done
done
