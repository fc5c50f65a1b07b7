#!/usr/bin/env python
"""bench.py - trimodal TriCoLo training samples/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[3] per-GPU shard = Tri(I+V): SparseCNNEncoder 32^3 voxels +
MVCNNEncoder 6x128^2 views + BiGRUEncoder 96 tokens, per-GPU batch 32 (global batch 32*N, weak scaling), NT-Xent over
the all-gathered global batch, backward, gradient all-reduce, Adam.  Synthetic batches are resident in HBM before the
timed region.  One step = forward + losses + backward + optimizer update; nothing is skipped or cached.

Extra objects on the JSON line: "roofline" (dominant kernel by measured time, timed live with HIP events on the
launch stream) and "cpu_baseline" (the oracle restatement of the same step on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0, "f16": 2500.0}      # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def build_net(args, device):
    from tricolo_amd import config as tcfg, ops
    from tricolo_amd.model.tricolo_net import TriCoLoNet
    ops.set_default_precision(args.precision)
    ov = ["data=synthetic", "model.text_encoder=BiGRUEncoder", "model.image_encoder=MVCNNEncoder",
          "model.voxel_encoder=SparseCNNEncoder", f"data.voxel_size={args.voxel_size}", f"data.num_views={args.num_views}",
          f"data.image_size={args.image_size}", "experiment_name=bench"]
    cfg = tcfg.compose(overrides=ov)
    torch.manual_seed(cfg.train_seed)                    # identical random-init weights on every rank
    net = TriCoLoNet(cfg).to(device)
    return net, cfg


def cpu_baseline(args, cfg):
    """Oracle (CPU restatement of the reference step) timed on this box's host cores on a bounded sample (<= ~30 s)."""
    from oracle import modules as om
    from tricolo_amd.data import synthetic as syn
    B = args.cpu_batch
    threads = max(1, min(os.cpu_count() or 1, args.cpu_threads))     # torch-CPU conv3d degrades when oversubscribed
    torch.set_num_threads(threads)
    torch.manual_seed(cfg.train_seed)
    ref = om.TriCoLoRef(om.BiGRURef(syn.DEFAULT_VOCAB, 512), om.MVCNNRef(512, 512, "resnet18", args.num_views),
                        om.SparseCNNRef(args.voxel_size, 32, 512, 512))
    opt = torch.optim.Adam(ref.parameters(), lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay)
    batch = syn.make_batch(B, voxel_size=args.voxel_size, num_views=args.num_views, image_size=args.image_size, seed=syn.BASE_SEED + 40)
    times, budget_end = [], time.perf_counter() + args.cpu_budget_s
    for i in range(1 + args.cpu_steps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref.training_step(batch)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > budget_end:
            break
    timed = sorted(times[1:]) if len(times) > 1 else times          # first step = warm-up unless the budget ran out
    med = timed[len(timed) // 2]
    return {"value": round(B / med, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{len(timed)} timed step(s) of the same Tri(I+V) fwd+bwd+Adam step at batch {B} on the torch-CPU oracle "
                      f"({threads} threads, median, {args.cpu_budget_s:.0f} s budget)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default=os.environ.get("TRICOLO_PRECISION", "bf16"), choices=["bf16", "bf16x3", "f16"])
    ap.add_argument("--per-gpu-batch", type=int, default=32)
    ap.add_argument("--voxel-size", type=int, default=32)
    ap.add_argument("--num-views", type=int, default=6)
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--resident-batches", type=int, default=2)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("TRICOLO_FORCE_DIST", "0") == "1"      # world-of-one run of the data-parallel code path
    if world > 1 or force_dist:
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from tricolo_amd import ops, parallel
    from tricolo_amd.data import synthetic as syn

    net, cfg = build_net(args, device)
    opt = net.configure_optimizers()
    if hasattr(opt, "prepare"):
        opt.prepare()
    B = args.per_gpu_batch
    batches = [syn.batch_to_device(syn.make_batch(B, voxel_size=args.voxel_size, num_views=args.num_views, image_size=args.image_size,
                                                  seed=syn.BASE_SEED + 4 + i, rank=rank), device)
               for i in range(args.resident_batches)]
    params = list(net.parameters())

    def step(batch):
        losses = parallel.dp_training_step(net, batch, opt)
        return losses["train_loss/total_loss"]

    # ---- HIP-graph capture of the step.  N = 1: one graph.  N > 1: three graphs around the two eager collectives
    # (parallel.GraphedDPStep); any capture failure falls back to the eager step.
    graphs = None
    dp_graph = (world > 1 or force_dist) and not args.no_graph
    if dp_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for i in range(2):
                    step(batches[i % len(batches)])                  # also creates the RCCL communicator before any capture
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graphs = [parallel.GraphedDPStep(net, opt, b) for b in batches]
            torch.cuda.synchronize()
        except Exception as e:                      # noqa: BLE001
            if rank == 0:
                print(f"[bench] graph-split DP step unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()
    use_graph = (not args.no_graph) and not dp_graph
    if use_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for i in range(2):
                    step(batches[i % len(batches)])
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graphs = []
            for b in batches:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    loss_static = step(b)
                graphs.append((g, loss_static))
            torch.cuda.synchronize()
        except Exception as e:                      # noqa: BLE001
            if rank == 0:
                print(f"[bench] HIP-graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = None
            torch.cuda.synchronize()

    def run(i):
        if graphs is not None:
            item = graphs[i % len(graphs)]
            if dp_graph:
                return item.replay()
            g, l = item
            g.replay()
            return l
        return step(batches[i % len(batches)])

    # clocks / caches settle over the first few dozen replays (measured: 4.7 ms per step averaged over steps 6-25 of a fresh
    # process, 4.45 ms over steps 6-45); pre-roll untimed steps so that the W + K steps below see the steady state
    for i in range(40 if graphs is not None else 5):
        run(i)
    for i in range(args.warmup):
        loss = run(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = run(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss.item())
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline leg: per-kernel HIP-event timing of eager steps, dominant kernel by time.  Rank 0 records; at N > 1
    # EVERY rank runs the steps (they contain the collectives - a rank-0-only step would wait for its peers forever).
    roof = None
    if rank == 0 or world > 1:
        if rank == 0:
            ops.TIMER = ops.KernelTimer()
        nprof = 3
        # serialise the streams for this leg: a kernel's HIP-event duration must not include the other towers' kernels
        # sharing the GPU with it (the timed region above keeps towers / weight gradients on parallel streams)
        net.overlap_towers = False
        for side_name in ("_side", "_side_ds"):
            if getattr(net.image_encoder, side_name, None) is not None:
                getattr(net.image_encoder, side_name).enabled = False
        for i in range(nprof):
            # park the GPU on a spin kernel first so the host enqueues the whole step ahead of it: the HIP events then
            # bracket back-to-back kernel execution instead of host launch gaps (eager launches cost ~10 us each)
            torch.cuda._sleep(int(60e6))
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
    if rank == 0:
        agg = ops.TIMER.summary()
        ops.TIMER = None
        if agg:
            sym, d = max(agg.items(), key=lambda kv: kv[1]["ms"])
            avg_ms = d["ms"] / d["launches"]
            tflops = d["flops"] / (d["ms"] * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            roof = {"bound": "mfma", "kernel": sym, "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(tflops / peak, 4), "traffic": None, "launches_per_step": d["launches"] // nprof,
                    "avg_launch_ms": round(avg_ms, 4),
                    "all_kernels": {k: {"launches_per_step": v["launches"] // nprof, "ms_per_step": round(v["ms"] / nprof, 3),
                                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in agg.items()}}

    # HBM bytes per launch of that kernel from the committed PMC passes over this same command (PMC cannot run inside bench.py:
    # two separate `rocprofv3 --pmc` runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes; profiles/r1/pmc_step_traffic.json)
    if roof is not None:
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1", "pmc_step_traffic.json")) as f:
                pmc = json.load(f)["kernels"].get(roof["kernel"])
            if pmc and args.precision == "bf16":
                roof["traffic"] = pmc["hbm_read_bytes_per_launch"] + (pmc["hbm_write_bytes_per_launch"] or 0)
                roof["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r1/pmc_step_traffic.json)"
        except Exception:                           # noqa: BLE001
            pass

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:             # the CPU leg is an N = 1 artefact (task contract)
        cpu = cpu_baseline(args, cfg)

    if rank == 0:
        gb = B * world
        out = {
            "metric": "trimodal training samples/sec (32^3 voxel + 6-view)", "value": round(gb * args.steps / elapsed, 2),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "bf16x3 (hi/lo split, fp32-grade)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] per-GPU shard: Tri(I+V) SparseCNN {args.voxel_size}^3 + MVCNN "
                                   f"{args.num_views}x{args.image_size}^2 + BiGRU-96, fwd+bwd+Adam",
                       "global_batch": gb, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "hip_graph": (("3 graphs + eager collectives" if dp_graph else True) if graphs is not None else False), "final_loss": round(final_loss, 5)},
            "roofline": roof, "cpu_baseline": cpu,
        }
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; push it out first so that the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                           # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
