"""The bench.py output contract (driver-facing): checks the latest committed bench line under profiles/ for every key
and type the task statement requires, and that bench.py's argument parser accepts the driver's command line."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench_lines():
    """The newest round's committed bench lines: profiles/rNN_bench_<config>.json (one per BASELINE config)."""
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json"))
    assert files, "no committed bench line under profiles/"
    rnd = max(int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) for f in files)
    return {os.path.basename(f).split("_bench_")[1][:-5]: json.load(open(f)) for f in files if os.path.basename(f).startswith(f"r{rnd:02d}_")}


def test_all_four_baseline_configs_have_a_committed_line():
    lines = _latest_bench_lines()
    assert set(lines) == {"embed", "frontend", "finetune", "stream"}
    assert lines["embed"]["config"]["workload"].startswith("configs[2]") and lines["frontend"]["config"]["workload"].startswith("configs[1]")
    assert lines["finetune"]["config"]["workload"].startswith("configs[3]") and lines["stream"]["config"]["workload"].startswith("configs[4]")
    assert lines["stream"]["unit"] == "windows/s" and lines["stream"]["latency_ms_batch1"] > 0
    assert lines["frontend"]["roofline"]["bound"] == "hbm" and lines["embed"]["roofline"]["bound"] == "mfma"
    builds = {d["config"]["build"] for d in lines.values()}
    assert len(builds) == 1                                            # one build, one box, one run of tools/gpu/evidence.sh


import pytest


@pytest.mark.parametrize("name", ["embed", "frontend", "finetune", "stream"])
def test_committed_bench_line_has_the_contract_fields(name):
    d = _latest_bench_lines()[name]
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "kernel" in r and r["avg_launch_ms"] > 0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    # consistency: value = clips per step / step time
    clips = d["config"]["clips_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - clips / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_bench_cli_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
