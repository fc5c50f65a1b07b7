#!/usr/bin/env python
"""Build profiles/<round>/pmc_traffic_<mode>.json from the two PMC passes of tools/profile_round.sh.

    python tools/pmc_traffic.py gpurun_out/<tag>_pmc_fetch gpurun_out/<tag>_pmc_write f16 > profiles/r2/pmc_traffic_f16.json

Per kernel symbol (named as tricolo_amd.ops names them for bench.py's KernelTimer: template variants that differ only in
record / store mode are merged) the mean FETCH_SIZE and WRITE_SIZE per launch.  Counter unit KiB; FETCH_SIZE is doubled
(gfx950 tallies 128-byte read requests at 64 bytes: MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import re
import subprocess
import sys


def demangle(names):
    # c++filt does not know the _Float16 / __bf16 manglings DF16_ / DF16b: stand-ins that it does know, renamed afterwards
    subst = [n.replace("DF16_", "t").replace("DF16b", "s") for n in names]
    out = subprocess.run(["c++filt"], input="\n".join(subst), capture_output=True, text=True, check=True).stdout.splitlines()
    res = {}
    for n, d in zip(names, out):
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)
        d = re.sub(r"\bunsigned short\b", "f16" if "DF16_" in n else "bf16", d)
        d = re.sub(r"\bshort\b", "bf16", d)
        res[n] = d
    return res


def symbol(d):
    m = re.match(r"conv_halo_rows_kernel<(\w+), \w+, \w+, \d+>", d)
    if m:
        return f"conv_halo_rows_kernel<{m.group(1)}>"
    m = re.match(r"conv_halo2d_kernel<(\d+), \w+, (\w+)>", d)
    if m:
        return f"conv_halo2d_kernel<{m.group(1)}, {m.group(2)}>"
    m = re.match(r"conv_dma_kernel<(\d+), \d+, (\w+)>", d)
    if m:
        return f"conv_dma_kernel<{m.group(1)}, {m.group(2)}>"
    m = re.match(r"conv_wgrad_dma_kernel<(\d+), (\d+), (\w+), \d+, \d+>", d)     # (wave layout: 2 x 2 unless TRICOLO_WGRAD_WIDE=1)
    if m:
        return f"conv_wgrad_dma_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}>"
    m = re.match(r"conv_wgrad_krow_kernel<(\d+), (\w+)(?:, \d+)?>", d)                # (stride-1 / stride-2 forms under one name)
    if m:
        return f"conv_wgrad_krow_kernel<{m.group(1)}, {m.group(2)}>"
    m = re.match(r"conv_stem_wgrad_kernel<(\d+), (\w+), \w+>", d)                  # (with / without the folded BatchNorm apply pass)
    if m:
        return f"conv_stem_wgrad_kernel<{m.group(1)}, {m.group(2)}>"
    m = re.match(r"conv_c64_kernel<(\w+), \d+, \d+, \w+, \w+>", d)             # (brick shape, accumulate, BatchNorm-backward sums)
    if m:
        return f"conv_c64_kernel<{m.group(1)}>"
    m = re.match(r"conv_s2d_kernel<(\w+), \d+, \d+, \w+>", d)
    if m:
        return f"conv_s2d_kernel<{m.group(1)}>"
    m = re.match(r"conv_vox0_wgrad_kernel<(\w+), \d+, \d+>", d)
    if m:
        return f"conv_vox0_wgrad_kernel<{m.group(1)}>"
    m = re.match(r"conv_vox(\d)_kernel<(\w+), \d+, \d+>", d)
    if m:
        return f"conv_vox{m.group(1)}_kernel<{m.group(2)}>"
    return d


def per_kernel(directory, counter):
    f = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    names = demangle(sorted({r["Kernel_Name"] for r in rows}))
    for r in rows:
        k = symbol(names[r["Kernel_Name"]])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, mode = sys.argv[1:4]
    ft, fc = per_kernel(fetch_dir, "FETCH_SIZE")
    wt, wc = per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(ft, key=lambda k: -ft[k]):
        if k not in wt:
            continue
        kernels[k] = {"dispatches": fc[k], "fetch_bytes_per_launch": int(ft[k] / fc[k] * 1024 * 2),
                      "write_bytes_per_launch": int(wt[k] / wc[k] * 1024)}
    # bytes of one training step: every kernel's total over the run / the number of steps the run executed (= launches of the fused
    # Adam kernel, one per step)
    steps = max([v["dispatches"] for k, v in kernels.items() if k.startswith("adam_seg_kernel") or k.startswith("adam_kernel")] or [0])
    step_bytes = int(sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["dispatches"] for v in kernels.values()) / steps) if steps else None
    json.dump({"step_bytes": step_bytes, "steps_in_run": steps, "method": f"two rocprofv3 runs of `python3 bench.py --steps 3 --warmup 2 --precision {mode} --modes '' --no-cpu-baseline`, one with "
                         "--pmc FETCH_SIZE, one with --pmc WRITE_SIZE (the TCC block cannot hold both); per-kernel mean over all dispatches of "
                         "the run; counter unit KiB; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md "
                         "'HBM'); Infinity-Cache hits are included in both counters, so this is fabric traffic >= HBM traffic; built by "
                         "tools/pmc_traffic.py",
               "kernels": kernels}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
