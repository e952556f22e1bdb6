cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -6
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print({k:d[k] for k in ('value','ms_per_step','roofline','whole_step','cpu_baseline')})"; tail -2 gpurun_out/bench_final.err
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_final -o fin -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_final.log 2>&1
ls gpurun_out/prof_final | head -5
