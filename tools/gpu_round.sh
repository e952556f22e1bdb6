# Round-end evidence run on the GPU box: full GPU test suite, smoke, default bench line (with CPU baseline),
# rocprofv3 kernel stats of the same bench command, and the two PMC passes for HBM traffic.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print({k:d[k] for k in ('value','ms_per_step','roofline','whole_step','cpu_baseline')})"; tail -2 gpurun_out/bench_final.err
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_final -o fin -- python bench.py --no-cpu-baseline > gpurun_out/prof_final.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/pmc_fetch -o f -- python tools/one_fwd.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d gpurun_out/pmc_write -o w -- python tools/one_fwd.py > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write | head -20
