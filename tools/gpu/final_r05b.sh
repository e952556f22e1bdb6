# refresh of the round-5 evidence after FORWARD_CLIPS = 2048 (host-side change: the device sources and their hash are those of final_r05.sh): the full
# -m gpu suite, kernel stats + PMC passes (now incl. the 2048-clip handle) + the four bench lines, fine-tune variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r05_pytest_gpu.log | tail -2
bash tools/gpu/final.sh r05
R=r05 bash tools/gpu/stats_cfg.sh > gpurun_out/r05_stats_cfg.log 2>&1
for g in 48 1; do
  timeout 300 python bench.py --config finetune --batch 64 --ft-group $g --no-cpu-baseline > gpurun_out/r05_variant_finetune64_g$g.json 2> gpurun_out/r05_variant_finetune64_g$g.err
  python -c "
import json;d=json.load(open('gpurun_out/r05_variant_finetune64_g$g.json'));print('finetune batch 64, steps per forward $g:', d['value'], 'clips/s', d['ms_per_step'], 'ms per optimizer step')"
done
for g in 1 2 4 6; do
  timeout 300 python bench.py --config finetune --no-cpu-baseline --ft-group $g > gpurun_out/r05_variant_finetune_g$g.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/r05_variant_finetune_g$g.json'));print('finetune 512, steps per forward $g:', d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'])"
done
timeout 300 python bench.py --config finetune --no-cpu-baseline --steps 20 --warmup 5 | python -c "
import json,sys;d=json.load(sys.stdin);print('finetune, driver-style 20 / 5:', d['value'], d['ms_per_step'])"
timeout 300 python tools/finetune_group_profile.py 512 6 > gpurun_out/r05_finetune_group_profile.txt 2>&1; grep "B=512\|device" gpurun_out/r05_finetune_group_profile.txt
python - <<'PY'
import json
for c in ("embed","frontend","finetune","stream"):
    d=json.load(open(f"gpurun_out/final/r05_bench_{c}.json")); r=d["roofline"]
    print(c, d["value"], d["unit"], d["ms_per_step"], r["kernel"], r["frac"], r.get("whole_step_frac"), r.get("time_weighted_frac"), r["traffic"], d.get("latency_ms_batch1"), d.get("latency_ms_batch1_eager"), d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread"]["value"])
PY
