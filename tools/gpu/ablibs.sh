# same-call bench of several library builds:  gpurun -- 'bash tools/gpu/ablibs.sh lib1.so lib2.so ...'  (paths relative to the repo; "-" = the shipped library)
#   KFILTER=substring restricts the printed kernels (default pw_gemm)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ablibs; mkdir -p $O
i=0
for l in "$@"; do
  if [ "$l" = "-" ]; then unset MKWS_LIB; else export MKWS_LIB=$PWD/$l; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_$i.json 2> $O/bench_$i.err
  i=$((i+1))
done
python - "$@" <<'PY'
import json, os, sys
flt = os.environ.get("KFILTER", "pw_gemm")
for i, l in enumerate(sys.argv[1:]):
    try:
        d=json.load(open(f"gpurun_out/ablibs/bench_{i}.json"))
        print(l, d["value"], d["ms_per_step"], {k: x["ms_per_step"] for k, x in d["kernels"].items() if flt in k})
    except Exception as e: print(l, "failed", e)
PY
