#!/usr/bin/env python
"""Lists, for every kernel of a .hip file, the inner loops that contain memory operations together with the sequence of
global_load / ds_read / s_waitcnt / v_mfma runs in the loop body (from `hipcc -S`).  Flags the two patterns that cost this
project the most (profiles/r01_notes.md):
  DRAIN   an `s_waitcnt vmcnt(0)` (or lgkmcnt(0)) inside a loop that also issues global (LDS) loads: the register ring is
          drained every iteration
  SUNK    all global loads of the body come after its last MFMA: the reloads were sunk to the loop end, so the next
          iteration starts by waiting for the freshest load
No GPU needed.   python tools/isa_audit.py [file.hip] [kernel-name-substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_asm(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-Wno-unused-value", "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernels(lines):
    cur, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and cur is None:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if "s_endpgm" in ln:
                yield cur, body
                cur = None


def demangle(name):
    try:
        return subprocess.check_output(["c++filt", name], text=True).strip()
    except Exception:
        return name


def loops(body):
    """(start, end) line ranges: from a 'Loop Header' label to the backward branch to that label."""
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = i
    res = []
    for i, ln in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i and "Loop Header" in body[labels[m.group(1)]]:
            res.append((labels[m.group(1)], i))
    return res


def summarise(seg):
    runs, prev, n = [], None, 0
    for ln in seg:
        t = ln.split()
        if not t:
            continue
        op = t[0]
        if op == "s_waitcnt":
            key = " ".join(t[:3]) if len(t) > 2 and "cnt" in t[2] else " ".join(t[:2])
        elif op.startswith(("global_load", "ds_read", "v_mfma", "global_store", "ds_write", "s_barrier")):
            key = op
        else:
            continue
        if key == prev:
            n += 1
        else:
            if prev:
                runs.append(f"{prev} x{n}" if n > 1 else prev)
            prev, n = key, 1
    if prev:
        runs.append(f"{prev} x{n}" if n > 1 else prev)
    return runs


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "multilingual_kws_amd", "csrc", "mkws_embed.hip")
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(compile_asm(src)):
        dn = demangle(name)
        if filt and filt not in dn:
            continue
        shown = False
        for s, e in loops(body):
            seg = body[s:e + 1]
            ngl = sum("global_load" in x for x in seg)
            nds = sum("ds_read" in x for x in seg)
            nm = sum("v_mfma" in x for x in seg)
            if ngl + nds == 0 or nm == 0:
                continue
            flags = []
            if ngl and any(re.search(r"s_waitcnt.*vmcnt\(0\)", x) for x in seg):
                flags.append("DRAIN(vm)")
            if nds and any(re.search(r"s_waitcnt.*lgkmcnt\(0\)", x) for x in seg) and nds > 1:
                flags.append("drain(lgkm)")
            idx_m = [i for i, x in enumerate(seg) if "v_mfma" in x]
            idx_l = [i for i, x in enumerate(seg) if "global_load" in x]
            if idx_l and idx_m and min(idx_l) > max(idx_m):
                flags.append("SUNK")
            if not shown:
                print(f"\n== {dn[:110]}")
                shown = True
            print(f"  loop @{s}..{e}: {nm} mfma, {ngl} global loads, {nds} ds_reads  {' '.join(flags)}")
            print("     " + "; ".join(summarise(seg))[:900])


if __name__ == "__main__":
    main()
