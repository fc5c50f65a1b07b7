#!/usr/bin/env python
"""Scan the gfx950 ISA of the library's kernels for the hazard round 6 found in gru_bwd_kernel: a vector-memory STORE whose data (or
address) VGPRs are overwritten by a VALU instruction a few instructions later.  The compiler treats that as safe (the store has "read" its
operands at issue); inside the replayed training step about one replay in 300 stored the NEW value in lanes 48..63 of one wave
(`global_store_dword ..., v74` two instructions ahead of `v_pk_mul_f32 v[74:75], v[74:75], ...`).

    python tools/isa_store_hazard.py [--window 6] [files ...]      (default: every .hip under tricolo_amd/csrc; runs without a GPU)

Prints every store with a VALU / load write to one of its source VGPRs inside the window; exit code 1 if any packed-math (v_pk_*) writer
is found (the form that failed), 0 otherwise."""
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    """VGPR numbers named by one operand token: v12, v[12:15]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(path, window):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-w", path, "-o", out])
        lines = open(out).read().split("\n")
    hits = []
    kernel = "?"
    body = []                                            # (kernel, text) of instruction lines only
    for ln in lines:
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            kernel = m.group(1)
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        body.append((kernel, t.split(";")[0].strip()))
    for i, (k, t) in enumerate(body):
        if not re.match(r"(global|buffer|flat|scratch)_store|global_atomic|buffer_atomic", t):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")] if " " in t else []
        # the DATA operand only: global / flat / scratch stores name it second (vaddr, vdata, ...), buffer stores first (vdata, vaddr, ...).
        # Address VGPRs are rewritten right behind their store all over the library (and by every compiler): they are read at issue.
        di = 0 if t.startswith("buffer_") else 1
        src = regs(ops[di].split()[0]) if len(ops) > di and ops[di] else set()
        for j in range(i + 1, min(i + 1 + window, len(body))):
            k2, t2 = body[j]
            if k2 != k or re.match(r"s_barrier|s_endpgm|s_cbranch|s_branch", t2):
                break
            mn = t2.split()[0]
            if not re.match(r"v_|ds_read|ds_load|global_load|buffer_load", mn) or re.match(r"v_cmp|v_readfirstlane|v_readlane", mn):
                continue
            dst_tok = t2.split(None, 1)[1].split(",")[0].strip() if " " in t2 else ""
            dst = regs(dst_tok)
            if dst & src:
                hits.append((k, t, j - i, t2))
                break
    return hits


def main():
    args = sys.argv[1:]
    window = 6
    if "--window" in args:
        window = int(args[args.index("--window") + 1])
        del args[args.index("--window"):args.index("--window") + 2]
    files = args or sorted(glob.glob(os.path.join(REPO, "tricolo_amd", "csrc", "*.hip")))
    bad = 0
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:          # (the time is hipcc's: threads are enough)
        results = list(ex.map(lambda f: scan(f, window), files))
    for f, hits in zip(files, results):
        pk = [h for h in hits if h[3].startswith("v_pk_")]
        bad += len(pk)
        print(f"{os.path.basename(f)}: {len(hits)} store(s) with a source VGPR rewritten within {window} instructions ({len(pk)} by packed math)")
        for k, t, dist, t2 in hits[:40]:
            print(f"    {k[:48]:48s} {t:60s} +{dist}: {t2}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
