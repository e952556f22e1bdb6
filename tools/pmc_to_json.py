#!/usr/bin/env python
"""Turns the two rocprofv3 PMC passes of tools/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; separate runs, as
the TCC counters do not fit one pass) into profiles/pmc_traffic.json, keyed by the kernel labels bench.py
prints.  Per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of 16 B/lane coalesced reads, so it is doubled; WRITE_SIZE is taken as is (uncalibrated).
The value is the mean over the launches of that kernel in the LAST forward pass of tools/one_fwd.py."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def label(name):
    """'void mkws::pw_gemm_kernel<2, 4, false>(mkws::GemmArgs)' -> 'pw_gemm_kernel<2,4,false>' (bench.py label)."""
    m = re.search(r"mkws::(\w+)(<[^>]*>)?", name)
    if not m:
        return None
    base, targs = m.group(1), (m.group(2) or "").replace(" ", "")
    if base in ("se_expand_kernel",):
        targs = targs.replace(",false", "")
    if base == "frontend_clip_kernel":
        targs = "<float,4>" if "float" in targs else targs
    return base + targs


def read_counter(dirpath, counter):
    files = glob.glob(os.path.join(dirpath, "*counter_collection.csv"))
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirpath}")
    per = {}
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = label(r["Kernel_Name"])
            if k:
                per.setdefault(k, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return per


def main():
    fdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_fetch")
    wdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "pmc_write")
    launches = json.load(open(os.path.join(ROOT, "tools", "launches_per_forward.json"))) if os.path.exists(
        os.path.join(ROOT, "tools", "launches_per_forward.json")) else {}
    fetch, write = read_counter(fdir, "FETCH_SIZE"), read_counter(wdir, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        def last_forward_mean(rows):
            rows = sorted(rows)
            n = launches.get(k)
            if not n:      # one_fwd.py runs 4 identical forwards: the last quarter of the dispatches
                n = max(1, len(rows) // 4)
            vals = [v for _, v in rows[-n:]]
            return sum(vals) / len(vals)
        fk = last_forward_mean(fetch[k]) if k in fetch else 0.0
        wk = last_forward_mean(write[k]) if k in write else 0.0
        out[k] = {"fetch_size_kib": round(fk, 1), "write_size_kib": round(wk, 1),
                  "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024))}
    dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  {k}")


if __name__ == "__main__":
    main()
