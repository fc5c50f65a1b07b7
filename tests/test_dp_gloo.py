"""CPU, world_size = 2, gloo: the data-parallel exchange of tricolo_amd.parallel (fused embedding all-gather with
autograd, SUM gradient all-reduce) against the golden vectors of the 2-shard case produced by the reference classes
(tests/golden/dp2_tri.npz: per-shard encoders with local BatchNorm statistics + NT-Xent over the gathered global batch).
The towers here are the CPU oracle modules - the collective logic under test is device-agnostic product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from itertools import combinations

    from oracle import modules as om
    from oracle.recipe import fill_module, probe
    from tricolo_amd import parallel
    from tricolo_amd.data import synthetic as syn

    net = om.TriCoLoRef(om.BiGRURef(syn.DEFAULT_VOCAB, 512), om.MVCNNRef(512, 512, "resnet18", 6), om.SparseCNNRef(32, 32, 512, 512))
    fill_module(net)
    full = syn.make_batch(8, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4)
    sl = slice(4 * rank, 4 * rank + 4)
    keep = (full["voxels"]["locs"][:, 0] >= 4 * rank) & (full["voxels"]["locs"][:, 0] < 4 * rank + 4)
    locs = full["voxels"]["locs"][keep].clone()
    locs[:, 0] -= 4 * rank
    shard = {"model_id": full["model_id"][sl], "category": full["category"][sl], "tokens": full["tokens"][sl],
             "images": full["images"][sl], "voxels": {"locs": locs, "feats": full["voxels"]["feats"][keep]}}
    local = net(shard)
    glob = parallel.gather_embeddings(local)                      # product code under test
    assert all(v.shape[0] == 8 for v in glob.values())
    total, per_pair = 0, {}
    for a, b in combinations(glob.keys(), 2):
        l = om.nt_xent_ref(glob[a], glob[b], 0.1, 0.25)
        per_pair[f"{a[:-9]}_{b[:-9]}"] = l.item()
        total = total + l
    total.backward()
    parallel.allreduce_gradients(list(net.parameters()))          # product code under test
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    flat2 = flat.clone() / 2
    parallel.allreduce_flat(flat2)                                # SUM of two halves == the already-reduced gradient
    res = {"total": total.item(), "flat_ok": bool(torch.allclose(flat2, flat, rtol=1e-6, atol=1e-8))}
    for name, p in net.named_parameters():
        n, s = probe(p.grad)
        res[f"gradnorm/{name}"] = n
    # the overlapped two-stage backward (parallel.backward_overlapped): cut at the output of layer2 of the image tower, the early
    # parameters' gradient ranges all-reduced asynchronously while the part below the cut still runs - same reduced gradient
    enc = net.image_encoder
    late = [p for idx in (0, 1, 4, 5) for p in enc.net_1[idx].parameters()]
    split = parallel.BackwardSplit(net, late)
    hook = enc.net_1[5].register_forward_hook(lambda m, i, o: split.gate(o))      # the gate replaces layer2's output
    net.zero_grad(set_to_none=True)
    glob2 = parallel.gather_embeddings(net(shard))
    hook.remove()
    total2 = sum(om.nt_xent_ref(glob2[a], glob2[b], 0.1, 0.25) for a, b in combinations(glob2.keys(), 2))
    flat3 = parallel.backward_overlapped(total2, split, list(net.parameters()))
    res["overlap_ok"] = bool(torch.allclose(flat3, flat, rtol=1e-5, atol=1e-7))
    res["overlap_late_fraction"] = sum(p.numel() for p in late) / flat.numel()
    res.update({f"loss/{k}": v for k, v in per_pair.items()})
    for k, v in glob.items():
        res[f"emb/{k}"] = v.detach().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_gather_and_gradient_allreduce_match_reference(golden, tmp_path):
    g = golden("dp2_tri")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2))
    for r in (r0, r1):
        assert abs(float(r["total"]) - float(g["total_loss"])) < 2e-5                    # identical global loss on every rank
        assert bool(r["flat_ok"])
        assert bool(r["overlap_ok"]) and float(r["overlap_late_fraction"]) < 0.06       # two-bucket path == single bucket
        for k in ("text_image", "text_voxel", "image_voxel"):
            assert abs(float(r[f"loss/{k}"]) - float(g[f"loss/{k}"])) < 2e-5
        for k in ("text_features", "image_features", "voxel_features"):
            np.testing.assert_allclose(r[f"emb/{k}"], g[f"emb/{k}"], atol=2e-6)          # rank-major row order
    worst = 0.0
    for key in g:
        if key.startswith("gradnorm/"):
            ref = float(g[key])
            for r in (r0, r1):                                                            # SUM over ranks == d L_global / d theta
                worst = max(worst, abs(float(r[key]) - ref) / max(ref, 1e-6))
    assert worst < 2e-3, worst
