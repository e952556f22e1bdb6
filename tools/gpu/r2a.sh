set -x
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
for c in embed frontend finetune stream; do
  timeout 400 python bench.py --config $c > gpurun_out/r2a/bench_$c.json 2> gpurun_out/r2a/bench_$c.err; echo "bench $c rc=$?"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof -o embed -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r2a | head -30
cat gpurun_out/r2a/bench_embed.json | head -c 1500
