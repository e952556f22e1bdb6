# timing-build trace lines (WG_FILTER) + a short bench line with the per-kernel table
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; rm -rf $O; mkdir -p $O
bash tools/gpu/wgtrace.sh | tail -n +2
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/embed.json 2> $O/embed.err; echo "embed rc=$? $(python -c "
import json;d=json.load(open('$O/embed.json'));print(d['value'],d['ms_per_step'],d['roofline']['whole_step_frac']);
for k,v in d['kernels'].items(): print('  %-50s %.4f ms  %.3f'%(k,v['ms_per_step'],v['frac']))")"
