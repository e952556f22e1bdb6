"""The bench.py output contract (driver-facing): checks the latest committed bench line under profiles/ for every key
and type the task statement requires, and that bench.py's argument parser accepts the driver's command line."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")),
                   key=lambda p: [int(x) for x in re.findall(r"\d+", os.path.basename(p))])
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_committed_bench_line_has_the_contract_fields():
    d = _latest_bench()
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
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
