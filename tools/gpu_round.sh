cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -8
python bench.py --steps 50 --warmup 10 > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r01b.json')); print({k:d[k] for k in ('value','ms_per_step','roofline','whole_step','cpu_baseline')})"; tail -3 gpurun_out/bench_r01b.err
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_r01b -o r01b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench_b.log 2>&1
ls gpurun_out/prof_r01b | head
