"""CPU: the C-ABI library loads and exports exactly the symbols include/tricolo_hip.h declares; the ctypes signature
table covers all of them; no compute is launched (there is no GPU here)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(REPO, "include", "tricolo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tri_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from tricolo_amd import _C
    if not os.path.exists(_C.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _C.lib()
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in tricolo_hip.h but not exported"
        assert s in _C.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_C.SIGNATURES) == syms
    assert lib.tri_version() >= 1
    assert lib.tri_conv_kpad(27, 4) == 128 and lib.tri_conv_kpad(49, 4) == 224


def test_host_helpers_without_gpu():
    from tricolo_amd import _C, ops
    g = ops.ConvGeom(2, (32, 32, 32), 3, 4, 32, (3, 3, 3), 1, (1, 1, 1), (81, 3, 1))
    assert g.out_grid == (32, 32, 32) and g.kpad == 128 and g.M == 65536 and g.num_mtiles[0] == 512
    g2 = ops.ConvGeom(12, (1, 128, 128), 3, 4, 64, (1, 7, 7), 2, (0, 3, 3), (147, 1, 49))
    assert g2.out_grid == (1, 64, 64) and g2.flops == 2 * 12 * 64 * 64 * 49 * 3 * 64
    assert g2.wgrad_ws > 0
    # round 4: the kernel-row weight-gradient planner (host-side; the launch itself needs a GPU) - which family a layer's job joins,
    # its tiles per position split and its 64-position steps
    l1 = ops.ConvGeom(192, (1, 32, 32), 64, 64, 64, (1, 3, 3), 1, (0, 1, 1), (576, 1, 9))
    assert l1.wgrad_krow and l1.wgrad_group(2) == (5, 3, 3072)                 # 64-row tiles: 3 kernel rows x 1 channel chunk
    l4 = ops.ConvGeom(192, (1, 4, 4), 512, 512, 512, (1, 3, 3), 1, (0, 1, 1), (4608, 1, 9))
    assert l4.wgrad_krow and l4.wgrad_group(2) == (4, 96, 48)                  # 4 channel tiles x 3 rows x 8 chunks
    s2 = ops.ConvGeom(192, (1, 32, 32), 64, 64, 128, (1, 3, 3), 2, (0, 1, 1), (576, 1, 9))
    assert s2.wgrad_krow and s2.wgrad_group(2) == (4, 3, 768)                  # stride 2 rides in the stride-1 launch
    w56 = ops.ConvGeom(8, (1, 56, 56), 128, 128, 128, (1, 3, 3), 1, (0, 1, 1), (1152, 1, 9))
    assert w56.wgrad_group(2) == (4, 6, 8 * 56)                                # one 56-position row per step
    ds = ops.ConvGeom(192, (1, 32, 32), 64, 64, 128, (1, 1, 1), 2, (0, 0, 0), (64, 1, 1))
    assert not ds.wgrad_krow and ds.wgrad_group(2)[0] in (1, 2)                # 1x1 shortcuts stay on the im2col family
    assert l1.wgrad_group(0) == (0, 0, 0)                                      # fp32 storage: no grouped family
    import torch
    with pytest.raises(RuntimeError, match="no CPU"):
        _C.ptr(torch.zeros(3))


def test_product_never_imports_oracle():
    import ast
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "tricolo_amd")):
        for f in files:
            if f.endswith(".py"):
                tree = ast.parse(open(os.path.join(root, f)).read())
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom) and node.module:
                        names = [node.module]
                    bad += [(f, n) for n in names if n.split(".")[0] == "oracle"]
    assert not bad, bad


def test_bench_gpus_n_spawns_its_ranks_and_fails_cleanly_without_devices():
    """`python bench.py --gpus 2` without a launcher starts two rank processes itself (before anything touches the GPU) and, on a box
    that lacks the devices, every rank exits with a message instead of an assertion / a hang."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    import torch
    if torch.cuda.device_count() == 0:
        assert r.stderr.count("needs an MI355X") == 2, r.stderr[-2000:]          # one message per spawned rank
    elif torch.cuda.device_count() == 1:
        assert "need 2 devices" in r.stderr, r.stderr[-2000:]
