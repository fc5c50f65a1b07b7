#!/usr/bin/env python
"""Register / spill / LDS report of the kernels in one HIP source file (device-only gfx950 compile, metadata notes of the code
object; runs without a GPU):  python tools/kernel_regs.py tricolo_amd/csrc/conv_igemm.hip [name regex]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "."
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output",
                               "-w", "-c", src, "-o", co] + sys.argv[3:])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        def g(k):
            m = re.search(k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = g(r"\.name")
        if not re.search(filt, name):
            continue
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        print("vgpr %4s spill %3s sgpr %4s scratch %5s lds %6s  %s" % (g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"), g(r"\.sgpr_count"),
                                                                  g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size"), name[:120]))


if __name__ == "__main__":
    main()
