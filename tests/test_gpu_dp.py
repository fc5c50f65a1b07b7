"""World size 2 on ONE MI355X: two rank processes share cuda:0 and exchange through the gloo backend (parallel.py stages device tensors
through host memory under gloo).  This runs the PRODUCT data-parallel path - HIP towers, parallel.gather_embeddings,
dp_training_step + FusedAdam.step(reduce_fn=...), and GraphedDPStep with the row slice of rank 1 - against the golden vectors of the
2-shard case produced by the reference classes (tests/golden/dp2_tri.npz, oracle/make_golden.py): per-shard encoders with local
BatchNorm statistics, NT-Xent over the gathered global batch, SUM of the ranks' parameter gradients.  The RCCL form of the same code
(backend "nccl") needs two GPUs; a one-rank RCCL world is covered by test_gpu_modules.py::test_graph_split_dp_step_equals_eager_dp_step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(full, rank, per):
    sl = slice(per * rank, per * rank + per)
    keep = (full["voxels"]["locs"][:, 0] >= per * rank) & (full["voxels"]["locs"][:, 0] < per * rank + per)
    locs = full["voxels"]["locs"][keep].clone()
    locs[:, 0] -= per * rank
    return {"model_id": full["model_id"][sl], "category": full["category"][sl], "tokens": full["tokens"][sl],
            "images": full["images"][sl], "voxels": {"locs": locs, "feats": full["voxels"]["feats"][keep]}}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.recipe import fill_module, probe
    from tricolo_amd import config as tcfg, ops, parallel
    from tricolo_amd.data import synthetic as syn
    from tricolo_amd.model.tricolo_net import TriCoLoNet

    ops.set_default_precision("bf16x3")
    cfg = tcfg.compose(overrides=["data=synthetic", "model.text_encoder=BiGRUEncoder", "model.image_encoder=MVCNNEncoder",
                                  "model.voxel_encoder=SparseCNNEncoder", "data.voxel_size=32", "data.num_views=6", "data.image_size=128",
                                  "experiment_name=dp2"])

    def build():
        net = TriCoLoNet(cfg)
        fill_module(net)
        return net.to(dev)

    full = syn.make_batch(8, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4)      # the batch of dp2_tri.npz
    shard = syn.batch_to_device(_shard(full, rank, 4), dev)
    res = {}

    # ---- 1. the exchange steps by hand: gathered embeddings, identical global loss, SUM-reduced flat gradient
    net = build()
    opt = net.configure_optimizers()
    opt.prepare()
    glob = parallel.gather_embeddings(net(shard))
    losses = net._calculate_losses(glob, "train_loss")
    total = losses["train_loss/total_loss"]
    total.backward()
    flat = opt.flat_grad()
    parallel.allreduce_flat(flat)
    res["total"] = total.item()
    for k, v in losses.items():
        res[f"loss/{k}"] = v.item()
    for k, v in glob.items():
        res[f"emb/{k}"] = v.detach().cpu().numpy()
    off = 0
    names = {id(p): n for n, p in net.named_parameters()}
    for p in opt._params:
        n = p.numel()
        gn, gs = probe(flat[off:off + n].view(p.shape).cpu())
        res[f"gradnorm/{names[id(p)]}"] = gn
        res[f"gradsample/{names[id(p)]}"] = gs
        off += n
    res["flat_sum"] = float(flat.double().sum().item())

    # ---- 2. dp_training_step (FusedAdam.step(reduce_fn=allreduce_flat)) for three steps, eagerly
    net_a = build()
    opt_a = net_a.configure_optimizers()
    opt_a.prepare()
    eager = [parallel.dp_training_step(net_a, shard, opt_a)["train_loss/total_loss"].item() for _ in range(4)]
    res["eager"] = np.array(eager)
    res["param_sum_eager"] = float(opt_a._flat_p.double().sum().item())

    # ---- 3. GraphedDPStep: three HIP graphs around the two (host-staged) collectives; rank 1 slices rows [4, 8) of d loss / d z
    net_b = build()
    opt_b = net_b.configure_optimizers()
    opt_b.prepare()
    warm = [parallel.dp_training_step(net_b, shard, opt_b)["train_loss/total_loss"].item() for _ in range(2)]
    torch.cuda.synchronize()
    gstep = parallel.GraphedDPStep(net_b, opt_b, shard)
    graphed = warm + [gstep.replay().item() for _ in range(2)]
    torch.cuda.synchronize()
    res["graphed"] = np.array(graphed)
    res["param_sum_graphed"] = float(opt_b._flat_p.double().sum().item())
    res["param_maxdiff_graph_vs_eager"] = float((opt_b._flat_p - opt_a._flat_p).abs().max().item())

    # ---- 4. the two-bucket overlapped backward, eagerly, from the same weights
    net_c = build()
    opt_c = net_c.configure_optimizers()
    opt_c.prepare()
    split = parallel.BackwardSplit.for_net(net_c)
    over = [parallel.dp_training_step(net_c, shard, opt_c, split=split)["train_loss/total_loss"].item() for _ in range(4)]
    res["overlap"] = np.array(over)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_dp2_product_path_two_ranks_on_one_gpu(golden, tmp_path):
    import torch.multiprocessing as mp
    g = golden("dp2_tri")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2))
    for r in (r0, r1):
        assert abs(float(r["total"]) - float(g["total_loss"])) < 1e-3                     # identical global loss on every rank
        for k in ("text_image", "text_voxel", "image_voxel"):
            assert abs(float(r[f"loss/train_loss/{k}_loss"]) - float(g[f"loss/{k}"])) < 1e-3
        for k in ("text_features", "image_features", "voxel_features"):
            np.testing.assert_allclose(r[f"emb/{k}"], g[f"emb/{k}"], atol=2e-4)          # rank-major rows, both shards' towers
    # SUM over ranks == d L_global / d theta: bit-identical on both ranks, and the reference's gradient (towers' bounds of test_gpu_modules)
    assert float(r0["flat_sum"]) == float(r1["flat_sum"])
    worst = {"voxel": 0.0, "text": 0.0, "image": 0.0}
    for key in g:
        if key.startswith("gradnorm/"):
            ref = float(g[key])
            tower = key.split("/")[1].split("_")[0]
            for r in (r0, r1):
                worst[tower] = max(worst[tower], abs(float(r[key]) - ref) / max(ref, 1e-6))
    assert worst["voxel"] < 2e-3 and worst["text"] < 2e-3 and worst["image"] < 2e-2, worst
    # the step itself: same trajectory on both ranks, eager == graphed == two-bucket, loss goes down, parameters stay replicated
    np.testing.assert_array_equal(r0["eager"], r1["eager"])
    np.testing.assert_array_equal(r0["graphed"], r1["graphed"])
    assert abs(float(r0["eager"][0]) - float(g["total_loss"])) < 1e-3
    np.testing.assert_allclose(r0["graphed"], r0["eager"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(r0["overlap"], r0["eager"], rtol=2e-4, atol=2e-4)
    assert r0["eager"][-1] < r0["eager"][0]
    assert float(r0["param_sum_eager"]) == float(r1["param_sum_eager"])
    assert float(r0["param_sum_graphed"]) == float(r1["param_sum_graphed"])
    for r in (r0, r1):
        assert float(r["param_maxdiff_graph_vs_eager"]) < 5e-3            # Adam's first steps are sign-like: bounded, not bitwise


def _worker_sync(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.recipe import fill_module
    from tricolo_amd import config as tcfg, ops, parallel
    from tricolo_amd.data import synthetic as syn
    from tricolo_amd.model.tricolo_net import TriCoLoNet

    ops.set_default_precision("bf16x3")
    ops.set_sync_bn(True)
    cfg = tcfg.compose(overrides=["data=synthetic", "model.text_encoder=BiGRUEncoder", "model.image_encoder=MVCNNEncoder",
                                  "model.voxel_encoder=SparseCNNEncoder", "data.voxel_size=32", "data.num_views=6", "data.image_size=128",
                                  "experiment_name=syncbn"])
    net = TriCoLoNet(cfg)
    fill_module(net)
    net = net.to(dev)
    opt = net.configure_optimizers()
    full = syn.make_batch(8, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4)      # the batch of step_cfg4_tri.npz
    shard = syn.batch_to_device(_shard(full, rank, 4), dev)
    res = {"loss": []}
    for step in range(3):
        out = parallel.gather_embeddings(net(shard))
        if step == 0:
            for k, v in out.items():
                res[f"emb/{k}"] = v.detach().cpu().numpy()
        losses = net._calculate_losses(out, "train_loss")
        opt.zero_grad(set_to_none=True)
        losses["train_loss/total_loss"].backward()
        opt.step(reduce_fn=parallel.allreduce_flat)
        res["loss"].append(losses["train_loss/total_loss"].item())
    res["loss"] = np.array(res["loss"])
    res["running_mean_sum"] = float(sum(b.double().sum().item() for n, b in net.named_buffers() if n.endswith("running_mean")))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sync_bn_two_ranks_reproduce_the_single_process_global_batch(golden, tmp_path):
    """SURVEY 8e "BatchNorm caveat": with ops.set_sync_bn(True) every BatchNorm uses global-batch statistics (one all-reduce of the
    per-channel sums + count per layer, forward and backward), so TWO ranks with 4 samples each reproduce the reference's
    SINGLE-process run on all 8 samples - tests/golden/step_cfg4_tri.npz, produced by the real TriCoLoNet: embeddings of the whole
    batch, the step-0 loss and the losses after one and two Adam steps (gradient SUM over ranks = the global-batch gradient)."""
    import torch.multiprocessing as mp
    g = golden("step_cfg4_tri")
    mp.spawn(_worker_sync, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2))
    np.testing.assert_array_equal(r0["loss"], r1["loss"])
    assert float(r0["running_mean_sum"]) == float(r1["running_mean_sum"])            # both ranks track the same global statistics
    for k in ("text_features", "image_features", "voxel_features"):
        np.testing.assert_allclose(r0[f"emb/{k}"], g[f"emb/{k}"], atol=2e-4)
    assert abs(float(r0["loss"][0]) - float(g["step0/total_loss"])) < 1e-3
    assert abs(float(r0["loss"][1]) - float(g["step1/total_loss"])) < 1e-2            # (Adam's first steps amplify round-off: the bound of
    assert abs(float(r0["loss"][2]) - float(g["step2/total_loss"])) < 1e-2            #  test_training_steps_match_reference)
