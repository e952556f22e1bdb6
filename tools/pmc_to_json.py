#!/usr/bin/env python
"""Turns the rocprofv3 PMC passes of tools/gpu/evidence.sh into committed summaries keyed by the kernel labels bench.py
prints:

  profiles/pmc_traffic.json   FETCH_SIZE / WRITE_SIZE (separate passes: the TCC counters do not fit one) -> HBM bytes
                              per launch.  Per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
                              FETCH_SIZE reports half the bytes of 16 B/lane coalesced reads, so it is doubled;
                              WRITE_SIZE is taken as is (uncalibrated).  "_source_hash" ties the file to the build of
                              the device sources it was measured on (bench.py refuses a stale file).
  profiles/rNN_mfma_busy.json SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: fraction of the kernel's active
                              cycles the MFMA pipes were busy, per kernel.

The value of a kernel is the mean over its launches in the LAST forward pass of tools/one_fwd.py."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIMDS = 32 * 4          # per XCD: GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, so busy / (active_sum * 128) = busy / (cycles * 1024)


def label(name):
    """'void mkws::pw_gemm_kernel<2, 4, false>(mkws::GemmArgs)' -> 'pw_gemm_kernel<2,4,false>' (bench.py label)."""
    m = re.search(r"mkws::(\w+)(<[^>]*>)?", name)
    if not m:
        return None
    base, targs = m.group(1), (m.group(2) or "").replace(" ", "")
    if base == "se_expand_kernel":
        targs = targs.replace(",false", "")
    if base == "mbconv_mid_kernel":       # device template <KS,S,KCT,HT,WT,CEXP,CC,NTP,G,SEG,NTHR,WPE> -> profile label <KS,S,HT,WT,CEXP,CC,G,NTHR>
        a = targs.strip("<>").split(",")
        targs = "<" + ",".join(a[i] for i in (0, 1, 3, 4, 5, 6, 8, 10)) + ">"
    return base + targs


def read_counters(dirpath):
    files = glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirpath}")
    per = {}
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            k = label(r["Kernel_Name"])
            if k:
                per.setdefault(r["Counter_Name"], {}).setdefault(k, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return per


def last_forward_mean(rows):
    rows = sorted(rows)
    n = max(1, len(rows) // 4)          # one_fwd.py runs 4 identical forwards: the last quarter of the dispatches
    vals = [v for _, v in rows[-n:]]
    return sum(vals) / len(vals)


def main():
    base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "evidence")
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    import bench
    fetch = read_counters(os.path.join(base, "pmc_fetch"))["FETCH_SIZE"]
    write = read_counters(os.path.join(base, "pmc_write"))["WRITE_SIZE"]
    out = {"_source_hash": bench.source_hash(), "_workload": "tools/one_fwd.py: frontend + embedding forward, B = 1024"}
    for k in sorted(set(fetch) | set(write)):
        fk = last_forward_mean(fetch[k]) if k in fetch else 0.0
        wk = last_forward_mean(write[k]) if k in write else 0.0
        out[k] = {"fetch_size_kib": round(fk, 1), "write_size_kib": round(wk, 1), "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024))}
    # the handles of the fine-tune (512 clips) and streaming (256) configs launch other workgroup shapes and move other byte counts per
    # launch: extra passes with ONE_FWD_B=512 / 256, stored as "<label>@<batch>" (bench.py looks a kernel up under its config's batch)
    for bsz in (3072, 2048, 512, 256):
        fd, wd = os.path.join(base, f"pmc_fetch_{bsz}"), os.path.join(base, f"pmc_write_{bsz}")
        if not (os.path.isdir(fd) and os.path.isdir(wd)):
            continue
        f5, w5 = read_counters(fd)["FETCH_SIZE"], read_counters(wd)["WRITE_SIZE"]
        for k in sorted(set(f5) | set(w5)):
            fk = last_forward_mean(f5[k]) if k in f5 else 0.0
            wk = last_forward_mean(w5[k]) if k in w5 else 0.0
            out[f"{k}@{bsz}"] = {"fetch_size_kib": round(fk, 1), "write_size_kib": round(wk, 1), "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024))}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for k, v in sorted(((k, v) for k, v in out.items() if not k.startswith("_")), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  {k}")
    mdir = os.path.join(base, "pmc_mfma")
    if os.path.isdir(mdir):
        c = read_counters(mdir)
        busy, active = c.get("SQ_VALU_MFMA_BUSY_CYCLES", {}), c.get("GRBM_GUI_ACTIVE", {})
        res = {"_source_hash": bench.source_hash(), "_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 128): GRBM_GUI_ACTIVE is summed over the 8 XCDs, each with 128 SIMDs; last forward of tools/one_fwd.py",
               "_note": "profiled passes run at a lower clock than un-profiled ones; the ratio is clock-independent"}
        for k in sorted(busy):
            if k in active:
                b, a = last_forward_mean(busy[k]), last_forward_mean(active[k])
                res[k] = {"mfma_busy_cycles": round(b), "gui_active_cycles": round(a), "mfma_busy_frac": round(b / (a * SIMDS), 4) if a else None}
        json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_mfma_busy.json"), "w"), indent=1)
        for k, v in sorted(((k, v) for k, v in res.items() if not k.startswith("_")), key=lambda kv: -(kv[1]["mfma_busy_frac"] or 0)):
            print(f"  MFMA busy {v['mfma_busy_frac']}  {k}")


if __name__ == "__main__":
    main()
