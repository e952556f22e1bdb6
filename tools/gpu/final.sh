# One call for the round's evidence on the final build: kernel stats + PMC passes, conversion ON the box (so that bench.py sees
# traffic of this very build), then the four bench lines.   gpurun -- 'bash tools/gpu/final.sh r02'
cd $GRAFT_REPO_ROOT
R=${1:-r02}
SKIP_BENCH=1 bash tools/gpu/evidence.sh > gpurun_out/final_evidence.log 2>&1
python tools/pmc_to_json.py gpurun_out/evidence $R > gpurun_out/final_pmc.log 2>&1
mkdir -p gpurun_out/final
cp profiles/pmc_traffic.json profiles/${R}_mfma_busy.json gpurun_out/final/
cp gpurun_out/evidence/stats/s_kernel_stats.csv gpurun_out/final/${R}_kernel_stats.csv
bash tools/gpu/benchlines.sh
for c in embed frontend finetune stream; do cp gpurun_out/benchlines/bench_$c.json gpurun_out/final/${R}_bench_$c.json; done
