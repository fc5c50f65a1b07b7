"""Static check of the compiled kernels (no GPU needed: hipcc cross-compiles): no packed-fp32 VALU instruction may rewrite the data
register of a vector-memory store a few instructions behind it - the pattern that made gru_bwd_kernel store a scaled value in lanes
48..63 of one wave about once in 300 replays of the training step (round 6; tools/isa_store_hazard.py, csrc/Makefile's -packed-fp32-ops)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_packed_math_writer_behind_a_store():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "isa_store_hazard.py"), "--window", "8"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "retrieval.hip" in r.stdout and "conv_igemm.hip" in r.stdout            # every source was scanned


def test_library_is_built_without_packed_fp32_ops():
    mk = open(os.path.join(REPO, "tricolo_amd", "csrc", "Makefile")).read()
    assert "-packed-fp32-ops" in mk
