#!/usr/bin/env python
"""bench.py - trimodal TriCoLo training samples/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload).  Default = BASELINE.json configs[3] per-GPU shard: Tri(I+V) = SparseCNNEncoder 32^3 voxels +
MVCNNEncoder 6x128^2 views + BiGRUEncoder 96 tokens, per-GPU batch 32 (global batch 32*N, weak scaling), NT-Xent over
the all-gathered global batch, backward, gradient all-reduce, Adam.  `--config 2|3|5` selects the other GPU configs of
BASELINE.json (Bi(V) B=64, Bi(I) B=64, Tri 64^3 + 12x224^2 + CLIP-text B=64 per GPU).  Synthetic batches are resident
in HBM before the timed region.  One step = forward + losses + backward + optimizer update; nothing is skipped or cached.

Precision modes, ALL timed by the same invocation at N = 1 (one JSON line):
  * `value` / `ms_per_step` / `roofline`: the f16 mode - f16 activation storage and MFMA operands, fp32 accumulation,
    bf16x3 heads / GRU.  It is the fastest mode whose embeddings and losses stay within the north star's 1e-3 of the fp32
    reference (asserted by tests/test_gpu_modules.py::test_f16_mode_meets_the_1e3_parity_bound on configs 1, 3, 4, 5).
  * `modes.bf16x3`: fp32 storage, 3-product split operands (strict parity: ~1e-5 on the losses).
  * `modes.bf16`: bf16 storage and operands (BASELINE config 2's dtype; OUTSIDE the 1e-3 bound: ~1.6e-3 / 2e-3).
Each mode carries its own `roofline` (dominant conv kernel by measured time, HIP events on the launch stream).
`roofline_3dconv_fwd`: the five SubMConv3d forwards of the voxel tower timed by HIP events, with dense AND executed FLOPs
(active 128-site tiles counted on the device from the site masks) and level 0's HBM rate.
`cpu_baseline`: the torch-CPU oracle on the host cores (CPU model, core counts and the probed thread counts stated; `cores` = the
threads used), 2 warm-up + >= 5 timed steps, for the bench workload at the per-GPU batch (halved until it fits the time budget) and
for BASELINE config 1 (Bi(V), batch 8).
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_PEAK_TFLOPS = 2500.0          # dense bf16 / f16 MFMA peak of MI355X (MI355X_MICROARCH.md); bf16x3 is priced against it too
HBM_PEAK_GBS = 8000.0

# BASELINE.json configs -> (text, image, voxel, V, views, image size, per-GPU batch)
CONFIGS = {
    2: ("BiGRUEncoder", None, "SparseCNNEncoder", 32, 6, 128, 64),
    3: ("BiGRUEncoder", "MVCNNEncoder", None, 32, 6, 128, 64),
    4: ("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128, 32),
    5: ("CLIPTextEncoder", "MVCNNEncoder", "SparseCNNEncoder", 64, 12, 224, 64),
}
DTYPE_NOTE = {
    "f16": "f16 (f16 activation storage + f16 MFMA operands, fp32 accumulate; bf16x3 heads/GRU; within 1e-3 of the fp32 reference)",
    "bf16x3": "bf16x3 (fp32 storage, hi/lo split bf16 MFMA operands, fp32-grade)",
    "bf16": "bf16 (bf16 activation storage + bf16 MFMA operands, fp32 accumulate)",
}


def workload_name(a):
    towers = "+".join(n for n in (f"SparseCNN {a.voxel_size}^3" if a.voxel else None,
                                  f"MVCNN {a.num_views}x{a.image_size}^2" if a.image else None,
                                  "BiGRU-96" if a.text == "BiGRUEncoder" else "CLIP-text MLP") if n)
    kind = "Tri(I+V)" if (a.voxel and a.image) else ("Bi(V)" if a.voxel else "Bi(I)")
    return f"BASELINE configs[{a.config - 1}] per-GPU shard: {kind} {towers}, fwd+bwd+Adam"


def step_gflop_per_sample(a):
    """Algorithmic FLOPs of one training step per sample (fwd + dgrad + wgrad ~ 3 x forward; BASELINE.md section 2 / SURVEY 8d)."""
    fwd = 0.057 if a.text == "BiGRUEncoder" else 0.002
    if a.voxel:
        fwd += {32: 1.0192, 64: 8.1537}.get(a.voxel_size, 1.0192 * (a.voxel_size / 32) ** 3)
    if a.image:
        fwd += {128: 1.1843, 224: 3.628}.get(a.image_size, 1.1843 * (a.image_size / 128) ** 2) * a.num_views
    return 3.0 * fwd


def build_net(a, precision, device):
    from tricolo_amd import config as tcfg, ops
    from tricolo_amd.model.tricolo_net import TriCoLoNet
    ops.set_default_precision(precision)
    ov = ["data=synthetic", f"model.text_encoder={a.text}", f"model.image_encoder={a.image or 'null'}",
          f"model.voxel_encoder={a.voxel or 'null'}", f"data.voxel_size={a.voxel_size}", f"data.num_views={a.num_views}",
          f"data.image_size={a.image_size}", "experiment_name=bench"]
    cfg = tcfg.compose(overrides=ov)
    torch.manual_seed(cfg.train_seed)                    # identical random-init weights on every rank
    net = TriCoLoNet(cfg).to(device)
    if a.text == "CLIPTextEncoder":
        pass                                             # Dropout(0.1) stays in train mode: it is part of the reference's step
    return net, cfg


def make_batches(a, rank, device, n):
    from tricolo_amd.data import synthetic as syn
    return [syn.batch_to_device(syn.make_batch(a.per_gpu_batch, voxel_size=a.voxel_size if a.voxel else None,
                                               num_views=a.num_views if a.image else None, image_size=a.image_size,
                                               clip_text=a.text == "CLIPTextEncoder", seed=syn.BASE_SEED + a.config + i, rank=rank), device)
            for i in range(n)]


def oracle_step_time(text, image, voxel, V, nv, S, B, cfg, threads, warm, timed, budget_s, seed):
    """Median wall time of the CPU oracle's fwd + bwd + Adam step (the reference restated in torch-CPU ops)."""
    from oracle import modules as om
    from tricolo_amd.data import synthetic as syn
    torch.set_num_threads(threads)
    torch.manual_seed(cfg.train_seed)
    t = om.BiGRURef(syn.DEFAULT_VOCAB, 512) if text == "BiGRUEncoder" else om.CLIPTextRef(512)
    ref = om.TriCoLoRef(t, om.MVCNNRef(512, 512, "resnet18", nv) if image else None, om.SparseCNNRef(V, 32, 512, 512) if voxel else None)
    opt = torch.optim.Adam(ref.parameters(), lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay)
    batch = syn.make_batch(B, voxel_size=V if voxel else None, num_views=nv if image else None, image_size=S,
                           clip_text=text == "CLIPTextEncoder", seed=seed)
    times, end = [], time.perf_counter() + budget_s
    for i in range(warm + timed):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref.training_step(batch)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > end and len(times) > warm:
            break
    kept = sorted(times[warm:]) if len(times) > warm else sorted(times)
    return kept[len(kept) // 2], len(kept), max(0, min(warm, len(times) - len(kept)))


def host_cpu_info():
    """CPU model string, logical CPUs and physical cores of this box (the report states model and core count)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return {"model": model, "logical_cpus": os.cpu_count() or 1, "physical_cores": len(cores) or None, "usable_cpus": usable}


def cpu_baseline(a, cfg):
    """The oracle timed on this box's host cores: 2 warm-up + >= 5 timed steps, median, on the SAME per-GPU batch
    as the GPU line when a step fits the time budget (the batch is halved until 2 + 5 steps fit --cpu-budget-s, judged from one probe
    step), plus BASELINE config 1 (Bi(V) 32^3 + BiGRU, batch 8).  Threads: BASELINE.md asks for torch.set_num_threads(os.cpu_count());
    on the 256-logical-CPU hosts of this pool that setting is 30 x SLOWER than 16 threads (measured in round 5: 0.67 against 19.8
    samples/s - torch's intra-op pools oversubscribe), so the leg first probes the thread counts 16, 32, 64, physical cores, all usable
    CPUs on one config-1 step each (stopping once a count is 1.5 x slower than the best so far), runs on the FASTEST and reports
    every probe beside the host's CPU model and core counts: `cores` = the threads actually used.  --cpu-threads N forces a count."""
    from tricolo_amd.data import synthetic as syn
    info = host_cpu_info()
    probes = {}
    if a.cpu_threads > 0:
        threads = max(1, min(info["usable_cpus"], a.cpu_threads))
    else:
        cands = sorted({c for c in (16, 32, 64, info["physical_cores"] or 0, info["usable_cpus"]) if 0 < c <= info["usable_cpus"]} or {1})
        best = None
        for c in cands:
            tm, _, _ = oracle_step_time("BiGRUEncoder", None, "SparseCNNEncoder", 32, 6, 128, 8, cfg, c, 1, 1, 4.0, syn.BASE_SEED + 1)
            probes[str(c)] = round(8 / tm, 2)
            if best is None or tm < best[0]:
                best = (tm, c)
            elif tm > 1.5 * best[0]:
                break
        threads = best[1]
    name = workload_name(a).split(': ')[1]
    B = a.cpu_batch or a.per_gpu_batch
    # probe: one step at batch 4 sizes the batch (a step scales ~linearly in the batch)
    probe, _, _ = oracle_step_time(a.text, a.image, a.voxel, a.voxel_size, a.num_views, a.image_size, min(4, B), cfg, threads, 1, 1, 5.0,
                                   syn.BASE_SEED + 40)
    per_sample = probe / min(4, B)
    while B > 4 and per_sample * B * (2 + a.cpu_steps) > a.cpu_budget_s:
        B //= 2
    med, n, w = oracle_step_time(a.text, a.image, a.voxel, a.voxel_size, a.num_views, a.image_size, B, cfg, threads, 2, a.cpu_steps,
                                 a.cpu_budget_s, syn.BASE_SEED + 40)
    out = {"value": round(B / med, 3), "unit": "samples/s", "cores": threads, "kind": "port",
           "cpu_model": info["model"], "host_logical_cpus": info["logical_cpus"], "host_physical_cores": info["physical_cores"],
           "host_usable_cpus": info["usable_cpus"], "batch": B, "ms_per_step": round(med * 1e3, 1),
           "thread_probe_config1_samples_per_s": probes,
           "sample": f"{n} timed steps after {w} warm-up of the same {name} step at batch {B} on the torch-CPU oracle ({threads} threads - "
                     f"the fastest of the probed thread counts on this {info['logical_cpus']}-CPU host -, median; the GPU line runs per-GPU "
                     f"batch {a.per_gpu_batch})"}
    med1, n1, w1 = oracle_step_time("BiGRUEncoder", None, "SparseCNNEncoder", 32, 6, 128, 8, cfg, threads, 2, a.cpu_steps,
                                    a.cpu_budget_s / 3, syn.BASE_SEED + 1)
    out["config1"] = {"value": round(8 / med1, 3), "unit": "samples/s", "ms_per_step": round(med1 * 1e3, 2), "cores": threads,
                      "sample": f"BASELINE configs[0] Bi(V) 32^3 + BiGRU, batch 8: {n1} timed steps after {w1} warm-up, median"}
    return out


def voxel_fwd_roofline(net, batch, B, nrep=3):
    """The five SubMConv3d forwards (sparse_cnn.py:12,17,22,27,32) timed by HIP events on their launch stream, with dense
    and EXECUTED FLOPs = executed 128-row tiles x 128 x 27 x Cin x Cout x 2, the tile count taken on the device from the level's
    site mask: levels that run over the compact active-site list execute ceil(active / 128) tiles, split-K levels (site mask)
    every 128-site tile that holds an active site.  `active_row_flops` is what a perfectly
    row-compacted kernel would execute."""
    from tricolo_amd import ops
    enc = net.voxel_encoder
    vox = batch["voxels"]
    ops.TIMER = ops.KernelTimer()
    torch.cuda._sleep(int(20e6))
    ovh = ops.TIMER.calibrate()                                        # event-pair cost of an empty launch (as in the step's roofline leg)
    saved = None
    for _ in range(nrep):
        torch.cuda._sleep(int(20e6))                                   # park the GPU: events bracket kernels, not launch gaps
        _, saved = enc._forward_impl(vox["locs"], vox["feats"], B, save=True)
    torch.cuda.synchronize()
    raw = [(s, f, a.elapsed_time(b)) for (s, f, a, b) in ops.TIMER.records if s.startswith("conv_")]
    recs = [(s, f, max(e - ovh, 0.25 * e)) for (s, f, e) in raw]
    ops.TIMER = None
    per = len(recs) // nrep                        # 5 SubMConv3d launches first, then (64^3: 8 head sites) mlp[0] on the conv path
    assert per >= 5 and per * nrep == len(recs), len(recs)
    levels, tot_ms, tot_raw, tot_dense, tot_exec, tot_rows = [], 0.0, 0.0, 0, 0, 0
    V = enc.voxel_size
    for l in range(5):
        x, y, mask, count, co, pooled, rows, used_rows = saved["levels"][l]
        D = V >> l
        M = B * D ** 3
        cin, cout = enc.chans[l], enc.chans[l + 1]
        m = mask[:M]
        pad = (-M) % 128
        if pad:
            m = torch.cat([m, m.new_zeros(pad)])
        active = int(m.sum().item())
        # executed row tiles: the compact row list packs the active sites into ceil(active / 128) tiles; split-K levels (site
        # mask instead of a list) still run every 128-site tile that holds an active site
        tiles = (active + 127) // 128 if used_rows else int(m.view(-1, 128).any(dim=1).sum().item())
        exec_rows = tiles * 128
        brick = recs[l][0].startswith("conv_vox")
        spu = enc._geom(B, l).voxg_spu(False, 2) if recs[l][0].startswith("conv_voxg") else 0
        if spu:                                        # conv_voxg_kernel: 16-row MFMA tiles over each unit's ranked active sites
            per_unit = m[:M].view(B, -1).sum(dim=1).to(torch.int64)
            if B % spu:
                per_unit = torch.cat([per_unit, per_unit.new_zeros(spu - B % spu)])
            per_unit = per_unit.view(-1, spu).sum(dim=1)
            # what the kernel EXECUTES: passes of up to NRT tiles (NRT by the unit's dense sites), the last pass's tile count rounded up to
            # the next instantiated loop (2 / 4 / 6 / 8 / 10 / 12, capped by NRT): conv_voxg.hip chunk_loop
            nsites = spu * D ** 3
            nrt_max = 2 if nsites <= 32 else (4 if nsites <= 64 else (8 if nsites <= 128 else 12))
            tiles_u = (per_unit + 15) // 16
            full, rest = tiles_u // nrt_max, tiles_u % nrt_max
            rest_up = torch.where(rest > 0, torch.clamp(((rest + 1) // 2) * 2, max=nrt_max), rest)
            exec_rows = int(((full * nrt_max + rest_up) * 16).sum().item())
            tiles = (exec_rows + 127) // 128
        elif recs[l][0].startswith("conv_voxb"):       # conv_voxb_kernel: 16-row tiles over each 2 x 4 x D brick's ranked active sites
            mb = m[:M].view(B, D // 2, 2, D // 4, 4, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, 8 * D).sum(dim=1).to(torch.int64)
            tiles_b = (mb + 15) // 16                  # passes of 8 tiles, the last one on the next instantiated count (1 / 2 / 3 / 4 / 6 / 8)
            full, rest = tiles_b // 8, tiles_b % 8
            rest_up = torch.where(rest == 5, torch.full_like(rest, 6), torch.where(rest == 7, torch.full_like(rest, 8), rest))
            exec_rows = int(((full * 8 + rest_up) * 16).sum().item())
            tiles = (exec_rows + 127) // 128
        elif brick:                                    # brick kernels (conv_vox.hip) execute the 16-site x-runs that hold an active site
            exec_rows = 16 * int(m.view(-1, 16).any(dim=1).sum().item())
            tiles = (exec_rows + 127) // 128
        ms = sorted(recs[r * per + l][2] for r in range(nrep))[nrep // 2]
        ms_raw = sorted(raw[r * per + l][2] for r in range(nrep))[nrep // 2]
        dense = 2 * M * 27 * cin * cout
        execd = 2 * exec_rows * 27 * cin * cout
        rowf = 2 * active * 27 * cin * cout
        e = x.element_size()
        cs = 4 if cin == 3 else cin
        # algorithmic HBM bytes: every needed input row once + every written output row once.  Over a compact row list only the
        # active sites' rows are written and (up to the 3^3 halo, counted as one extra shell = x2) read; a dense pass moves all M rows
        hbm = (min(M, 2 * active) * cs + active * cout) * e if (used_rows or brick) else M * (cs + cout) * e
        levels.append({"level": l, "kernel": recs[l][0], "grid": D, "cin": cin, "cout": cout, "sites": M, "active_sites": active,
                       "tiles": (M + 127) // 128, "executed_tiles": tiles, "executed_rows": exec_rows, "compact_rows": bool(used_rows), "ms": round(ms, 4),
                       "ms_raw": round(ms_raw, 4), "dense_tflops": round(dense / ms / 1e9, 1),
                       "executed_tflops": round(execd / ms / 1e9, 1), "active_row_tflops": round(rowf / ms / 1e9, 1),
                       "algorithmic_hbm_gbs": round(hbm / ms / 1e6, 1)})
        tot_ms += ms; tot_raw += ms_raw; tot_dense += dense; tot_exec += execd; tot_rows += rowf
    # the same five launches BACK TO BACK: each level's conv forward captured 20 x into a HIP graph and replayed between one event pair
    # (no per-launch event cost, no host launch gaps: kernel + launch boundary), median of 3 replays
    tot_b2b = 0.0
    for l in range(5):
        x, y, mask, count, co, pooled, rows, used_rows = saved["levels"][l]
        g = enc._geom(B, l)
        packed = enc._packed[(l, False)]
        sel = dict(rows=rows) if used_rows else dict(row_mask=mask)
        out = torch.empty_like(y)

        def fn():
            ops.conv_fwd(x, g, packed, want_stats=True, out=out, **sel)
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        NREP = 20
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(NREP):
                fn()
        ts = []
        for _ in range(3):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            gr.replay()
            b_.record()
            torch.cuda.synchronize()
            ts.append(a_.elapsed_time(b_) / NREP)
        levels[l]["ms_back_to_back"] = round(sorted(ts)[1], 4)
        tot_b2b += sorted(ts)[1]
        del gr
    l0 = levels[0]
    return {"bound": "mfma", "what": "five SubMConv3d forwards of the voxel tower, per-GPU batch %d, %d^3" % (B, V),
            "ms_back_to_back": round(tot_b2b, 4),
            "achieved_back_to_back": round(tot_exec / tot_b2b / 1e9, 2), "frac_back_to_back": round(tot_exec / tot_b2b / 1e9 / MFMA_PEAK_TFLOPS, 4),
            # constant work across kernel generations: the FLOPs of the active rows alone (what no kernel can skip).  `executed_flops` shrinks when a
            # kernel stops multiplying inactive rows of a tile / run (round 5: level 1 executes 1.15 x its active rows instead of 2.2 x), so the
            # executed-FLOP fraction can stay flat while the launches get faster; these two move only with time
            "frac_active_rows_raw_events": round(tot_rows / tot_raw / 1e9 / MFMA_PEAK_TFLOPS, 4),
            "frac_active_rows_back_to_back": round(tot_rows / tot_b2b / 1e9 / MFMA_PEAK_TFLOPS, 4),
            "timing_back_to_back": "each level's launch replayed 20 x back to back from a HIP graph between ONE event pair (kernel + launch boundary, "
                                   "no per-launch event cost); `frac` / `achieved` stay on the single-launch RAW event times as in round 4",
            "ms": round(tot_ms, 4), "ms_raw": round(tot_raw, 4),
            "timing": f"HIP events around each launch, median of {nrep}; the event-pair cost of an empty launch measured in the same leg "
                      f"({ovh * 1e3:.1f} us) is subtracted per launch (ms_raw = unsubtracted)",
            "dense_flops": tot_dense, "executed_flops": tot_exec, "active_row_flops": tot_rows,
            # `achieved` / `frac` are priced on the RAW event times (an upper bound of the kernels' own durations: event
            # processing + dispatch gap included); the overhead-subtracted figure is kept beside it as *_minus_event_overhead
            "achieved": round(tot_exec / tot_raw / 1e9, 2), "achieved_dense_equivalent": round(tot_dense / tot_raw / 1e9, 2),
            "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tot_exec / tot_raw / 1e9 / MFMA_PEAK_TFLOPS, 4),
            "frac_dense_equivalent": round(tot_dense / tot_raw / 1e9 / MFMA_PEAK_TFLOPS, 4),
            "frac_minus_event_overhead": round(tot_exec / tot_ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
            "rocprof": "kernel durations of the same five launches: newest profiles/r*/kernel_stats_<mode>.csv (conv_vox0_kernel, conv_voxb_kernel, "
                       "conv_voxg_kernel / conv_dma_kernel / conv_igemm_kernel rows), and that round's kernel_stats_voxel_fwd.csv / voxel_fwd.txt",
            "level0_hbm": {"bound": "hbm", "achieved": l0["algorithmic_hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(l0["algorithmic_hbm_gbs"] / HBM_PEAK_GBS, 4),
                           "note": "level 0 (3 -> 32 channels, 74 FLOP/B) is HBM-bound: bytes = needed input rows once + written output "
                                   "rows once (active sites only when the level runs over the compact row list)"},
            "levels": levels}


def run_mode(a, precision, world, rank, device, want_voxel_roofline):
    """Build the net in one precision mode, capture the step, time K steps, then the per-kernel roofline leg."""
    from tricolo_amd import ops, parallel
    net, cfg = build_net(a, precision, device)
    opt = net.configure_optimizers()
    if hasattr(opt, "prepare"):
        opt.prepare()
    B = a.per_gpu_batch
    batches = make_batches(a, rank, device, a.resident_batches)
    # initial parameters + BatchNorm buffers (restored before the timed windows)
    snap = (opt._flat_p.clone(), [b.clone() for b in net.buffers()]) if getattr(opt, "_flat_p", None) is not None else None
    force_dist = os.environ.get("TRICOLO_FORCE_DIST", "0") == "1"

    # N > 1, TRICOLO_DP_OVERLAP=1: two-stage backward (parallel.BackwardSplit) - the all-reduce of the early two thirds of the
    # gradient bytes runs under the second stage.  Off by default: on ONE GPU the cut costs 0.45 ms of GPU time per step (two
    # graphs instead of one: 1.78 + 1.51 ms against 2.85 ms, profiles/r2/README.md), about what the hidden part of the 74 MB
    # all-reduce is expected to take at 8 GPUs - without a multi-GPU box to measure on, the simpler single-bucket step stays.
    split = None
    # default (round 6, DESIGN.md section 6 "scaling model"): the two-bucket backward costs +0.45 ms of GPU time per step (four graphs, a
    # second pack) and hides 68 % of the gradient all-reduce - it wins only where the exposed all-reduce is longer than 0.45 / 0.68 =
    # 0.66 ms: by the per-link model that is N = 2 (one xGMI link between the pair: 74 MB each way at ~76 GB/s = 0.97 ms), not N = 4 / 8
    ov_default = "1" if (world == 2 and not force_dist) else "0"
    if (world > 1 or force_dist) and os.environ.get("TRICOLO_DP_OVERLAP", ov_default) == "1":
        split = parallel.BackwardSplit.for_net(net)

    def step(batch):
        return parallel.dp_training_step(net, batch, opt, split=split)["train_loss/total_loss"]

    def warm_eager():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(2):
                step(batches[i % len(batches)])              # also creates the RCCL communicator before any capture
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    # ---- HIP-graph capture of the step.  N = 1: one graph.  N > 1: three graphs around the two eager collectives
    # (parallel.GraphedDPStep).  A capture failure falls back to the eager step and is recorded in config.hip_graph.
    graphs, graph_note = None, False
    dp_graph = (world > 1 or force_dist) and not a.no_graph
    if not a.no_graph:
        try:
            warm_eager()
            if dp_graph:
                graphs = [parallel.GraphedDPStep(net, opt, b, split=split) for b in batches]
                graph_note = ("4 graphs + eager collectives, early-bucket all-reduce under the lower backward" if split is not None
                              else "3 graphs + eager collectives")
            else:
                graphs = []
                pool = torch.cuda.graph_pool_handle()            # one pool: the graphs replay one after the other and reuse each other's activations
                for b in batches:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool):
                        loss_static = step(b)
                    graphs.append((g, loss_static))
                graph_note = True
            torch.cuda.synchronize()
        except Exception as e:                      # noqa: BLE001
            if rank == 0:
                print(f"[bench] HIP-graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs, graph_note = None, f"capture failed: {type(e).__name__}"
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

    # clocks / caches settle over the first few dozen replays; pre-roll untimed steps so the W + K steps see the steady state
    for i in range(a.preroll if graphs is not None else 5):
        run(i)
    # the timed steps are steps W .. W + 3K of a run from the INITIAL weights, not those of a net that has memorised its resident batches
    # (after ~250 replays of two batches the routing the timed step saw - ReLU masks, view arg-max - was an over-fitted net's)
    if snap is not None:
        with torch.no_grad():
            opt._flat_p.copy_(snap[0]); opt._flat_m.zero_(); opt._flat_v.zero_(); opt._step_dev.zero_()
            for bufr, b0 in zip(net.buffers(), snap[1]):
                bufr.copy_(b0)
    for i in range(a.warmup):
        loss = run(i)
    # three back-to-back windows of EXACTLY K steps, each bracketed by barrier + synchronize, max over ranks per window; `value` is the
    # MEDIAN window (a 66 ms region is at the mercy of one host hiccup), min / max beside it
    windows = []
    for w in range(a.windows):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            loss = run(w * a.steps + i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        windows.append(el)
    elapsed = sorted(windows)[len(windows) // 2]
    final_loss = float(loss.item())
    nonfinite = ({"skipped_steps": opt.skipped_steps(), "skipped_elements": opt.nonfinite_skipped()} if hasattr(opt, "skipped_steps") else None)

    # ---- data-parallel breakdown (N > 1, or TRICOLO_FORCE_DIST=1 in a world of one): the exposed time of the two exchange steps and
    # of the three graph segments, HIP events on this rank's stream, median of 10 replays behind the timed windows; every rank reports
    # (rank 0 gathers the lines) together with the number of ranks the RCCL communicator really spans.
    dp_info = None
    have_graphs = dp_graph and graphs is not None
    if dp_graph and world > 1:                       # the leg below holds collectives: every rank must take the same branch
        ok = torch.tensor([1.0 if have_graphs else 0.0], dtype=torch.float32, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        have_graphs = have_graphs and float(ok.item()) > 0.5
    if have_graphs:
        seen = torch.ones((1,), dtype=torch.float32, device=device)
        parallel.all_reduce_sum(seen)
        evs = []
        for i in range(10):
            _, ev = graphs[i % len(graphs)].replay_timed()
            evs.append(ev)
        torch.cuda.synchronize()
        ph = [parallel.GraphedDPStep.phase_ms(ev) for ev in evs]
        med = {k: round(sorted(p_[k] for p_ in ph)[len(ph) // 2], 4) for k in ph[0]}
        mine = {"rank": rank, "backend": dist.get_backend(), "ranks_seen": int(round(float(seen.item()))), "phase_ms_median": med,
                "all_gather_bytes_per_rank": int(graphs[0].packed.numel() * 4), "all_reduce_bytes": int(graphs[0].flat.numel() * 4)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        dp_info = {"what": "HIP events around the phases of GraphedDPStep.replay on each rank's stream (10 replays, median): all_gather / "
                           "all_reduce are the EXPOSED times of the two RCCL collectives", "overlap_split": split is not None,
                   "per_rank": gathered}
        print(f"[bench] rank {rank}: {dist.get_backend()} ranks seen = {mine['ranks_seen']} (WORLD_SIZE {world}); exposed all-gather "
              f"{med['all_gather']:.3f} ms, all-reduce {med['all_reduce']:.3f} ms", file=sys.stderr, flush=True)

    # ---- roofline leg: per-kernel HIP-event timing of eager steps, dominant kernel by time.  Rank 0 records; at N > 1
    # EVERY rank runs the steps (they contain the collectives - a rank-0-only step would wait for its peers forever).
    roof, vox_roof = None, None
    if rank == 0:
        ops.TIMER = ops.KernelTimer()
    nprof = 6                          # (eager steps of the leg: 72 launches of the graded family instead of 36 - the run-to-run spread of `frac` was +-3 %)
    # serialise the streams for this leg: a kernel's HIP-event duration must not include the other towers' kernels
    # sharing the GPU with it (the timed region above keeps towers on parallel streams)
    net.overlap_towers = False
    for side_name in ("_side", "_side_ds", "_side_prep"):
        if net.image_encoder is not None and getattr(net.image_encoder, side_name, None) is not None:
            getattr(net.image_encoder, side_name).enabled = False
    if rank == 0:
        torch.cuda._sleep(int(20e6))
        ops.TIMER.calibrate()
    for i in range(nprof):
        # park the GPU on a spin kernel first so the host enqueues the whole step ahead of it: the HIP events then
        # bracket back-to-back kernel execution instead of host launch gaps (eager launches cost ~10 us each)
        torch.cuda._sleep(int(60e6))
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    if rank == 0:
        agg = {k: v for k, v in ops.TIMER.summary().items() if k.startswith("conv_")}
        ev_ovh = ops.TIMER.overhead_ms
        ops.TIMER = None
        if agg:
            # dominant kernel = the FAMILY (all template instantiations of one __global__ function: tile shapes, record / store variants,
            # pipeline depths) with the most measured time; per-instantiation lines stay in all_kernels
            fam = {}
            for k, v in agg.items():
                f = fam.setdefault(k.split("<")[0], {"launches": 0, "ms": 0.0, "ms_raw": 0.0, "flops": 0, "flops_dense": 0, "variants": []})
                f["launches"] += v["launches"]; f["ms"] += v["ms"]; f["ms_raw"] += v["ms_raw"]; f["flops"] += v["flops"]
                f["flops_dense"] += v["flops_dense"]
                f["variants"].append(k)
            fname, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
            sym = f"{fname}<*> ({len(d['variants'])} instantiation(s): {', '.join(sorted(d['variants']))})"
            tflops = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": sym, "achieved": round(tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / MFMA_PEAK_TFLOPS, 4),
                    "frac_raw_events": round(d["flops"] / (d["ms_raw"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),      # no event-overhead subtraction
                    "traffic": None, "launches_per_step": d["launches"] // nprof,
                    "avg_launch_ms": round(d["ms"] / d["launches"], 4), "avg_launch_ms_raw": round(d["ms_raw"] / d["launches"], 4),
                    "timing": "HIP events around every launch of eager, stream-serialised steps; the event-pair overhead of an empty "
                              f"kernel measured in the same leg ({ev_ovh * 1e3:.1f} us) is subtracted per launch (raw value beside it); "
                              "compare avg_launch_ms with the rocprofv3 AverageNs of the newest profiles/r*/kernel_stats_<mode>.csv",
                    "flops_basis": "EXECUTED FLOPs: 2*M*taps*Cin*Cout of dense launches, 2*(active rows)*taps*Cin*Cout of launches over a compact "
                                   "row list / site mask (device-side count read after the leg); achieved_dense_equivalent prices the same time "
                                   "on every site",
                    "achieved_dense_equivalent": round(d["flops_dense"] / (d["ms"] * 1e-3) / 1e12, 2),
                    "families": {k: {"launches_per_step": v["launches"] // nprof, "ms_per_step": round(v["ms"] / nprof, 3),
                                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in fam.items()},
                    "all_kernels": {k: {"launches_per_step": v["launches"] // nprof, "ms_per_step": round(v["ms"] / nprof, 3),
                                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in agg.items()}}
            # HBM bytes per launch of that kernel from the committed PMC passes over this same command (PMC cannot run inside
            # bench.py: separate `rocprofv3 --pmc` runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes)
            import glob
            pmc_docs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", f"pmc_traffic_{precision}.json")),
                              key=lambda f: int("".join(c for c in os.path.basename(os.path.dirname(f)) if c.isdigit()) or 0), reverse=True)
            for path in pmc_docs:                    # newest round first
                rel = os.path.relpath(path, REPO)
                try:
                    with open(os.path.join(REPO, rel)) as f:
                        doc = json.load(f)
                    hit = {k: v for k, v in doc["kernels"].items() if k.split("<")[0] == fname}
                    n = sum(v["dispatches"] for v in hit.values())
                    if n:
                        roof["traffic"] = int(sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["dispatches"] for v in hit.values()) / n)
                        roof["traffic_unit"] = (f"fabric bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, mean over the family's launches; {rel}; "
                                                "PMC cannot run inside bench.py)")
                        roof["step_hbm_bytes"] = doc.get("step_bytes")
                        break
                except Exception:                   # noqa: BLE001  no committed PMC pass for this mode: traffic stays null
                    continue
        if want_voxel_roofline and net.voxel_encoder is not None and "locs" in batches[0]["voxels"]:
            vox_roof = voxel_fwd_roofline(net, batches[0], B)
    gflop_per_sample = step_gflop_per_sample(a)
    res = {"value": round(B * world * a.steps / elapsed, 2), "ms_per_step": round(elapsed / a.steps * 1e3, 3),
           "ms_per_step_windows": {"median": round(elapsed / a.steps * 1e3, 3), "min": round(min(windows) / a.steps * 1e3, 3),
                                   "max": round(max(windows) / a.steps * 1e3, 3), "windows": len(windows), "steps_per_window": a.steps},
           "step_tflops": round(gflop_per_sample * B * world * a.steps / elapsed / 1e3, 1),
           "step_tflops_note": f"algorithmic {gflop_per_sample:.2f} GFLOP per sample (3 x forward: SURVEY 8d / BASELINE.md section 2) over the median window; "
                               f"{100 * gflop_per_sample * B * a.steps / elapsed / 1e3 / MFMA_PEAK_TFLOPS / 1:.1f} % of one GPU's dense 16-bit MFMA peak per GPU",
           "dtype": DTYPE_NOTE[precision], "hip_graph": graph_note if graphs is not None else (graph_note or False),
           "final_loss": round(final_loss, 5), "nonfinite_guard": nonfinite, "data_parallel": dp_info, "roofline": roof}
    del graphs, net, opt, batches
    # per-stream scratch arenas of this mode's (now dead) streams: hand them back before the next mode allocates - a bf16x3 leg run
    # after an f16 leg in the same process was 15 % slower than on its own until they were released
    ops._wgrad_ws.clear()
    ops.WgradBatch._arenas.clear()
    gc.collect()
    torch.cuda.empty_cache()
    return res, vox_roof, cfg


def run_mode_in_child(a, mode):
    """`bench.py --precision <mode> --modes '' --no-cpu-baseline` with this run's workload flags in a child process (a child, not an
    exec: this process has initialised the GPU); returns the same per-mode object run_mode() does."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--precision", mode, "--modes", "", "--no-cpu-baseline", "--steps", str(a.steps),
           "--warmup", str(a.warmup), "--config", str(a.config), "--per-gpu-batch", str(a.per_gpu_batch),
           "--resident-batches", str(a.resident_batches), "--preroll", str(a.preroll)]
    for flag, val in (("--voxel-size", a.voxel_size), ("--num-views", a.num_views), ("--image-size", a.image_size)):
        if val is not None:
            cmd += [flag, str(val)]
    if a.no_graph:
        cmd.append("--no-graph")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"bench.py child for mode {mode} failed (rc {r.returncode}):\n{r.stderr[-2000:]}")
    d = json.loads(lines[-1])
    return {"value": d["value"], "ms_per_step": d["ms_per_step"], "ms_per_step_windows": d.get("ms_per_step_windows"), "step_tflops": d.get("step_tflops"),
            "dtype": d["dtype"], "hip_graph": d["config"]["hip_graph"], "final_loss": d["config"]["final_loss"], "roofline": d["roofline"],
            "nonfinite_guard": d.get("nonfinite_guard"), "process": "child"}


def voxel_fwd_config5_in_child():
    """The five SubMConv3d forwards at BASELINE config 5's per-GPU shape (64^3 grids, batch 64) - the shape where north_star's "40 % of
    the MFMA roofline on the 3D-conv forward" is not below five launch latencies - timed by tools/voxel_fwd_bench.py in a child process
    (its own allocator state; ~20 s).  Same object as `roofline_3dconv_fwd`; the PRIMARY figure is `frac_active_rows_raw_events`:
    FLOPs of the active rows alone (constant across kernel generations) over the raw single-launch event times."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        js = os.path.join(d, "v.json")
        cmd = [sys.executable, os.path.join(REPO, "tools", "voxel_fwd_bench.py"), "--modes", "f16", "--shapes", "64x64", "--json", js]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not os.path.exists(js):
            raise RuntimeError(f"voxel_fwd_bench.py child failed (rc {r.returncode}):\n{r.stderr[-1500:]}")
        with open(js) as f:
            doc = json.load(f)
    res = next(iter(doc.values()))
    res["primary"] = "frac_active_rows_raw_events"
    res["frac_executed_raw_events"] = res["frac"]
    res["process"] = "child (tools/voxel_fwd_bench.py --modes f16 --shapes 64x64)"
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default=os.environ.get("TRICOLO_PRECISION", "f16"), choices=["bf16", "bf16x3", "f16"],
                    help="mode quoted as `value` (default f16: fastest mode inside the 1e-3 parity bound)")
    ap.add_argument("--modes", default=None, help="comma list of additional modes timed into `modes` (default at N=1: the other two)")
    ap.add_argument("--config", type=int, default=4, choices=[2, 3, 4, 5], help="BASELINE.json config number (1-based)")
    ap.add_argument("--per-gpu-batch", type=int, default=None)
    ap.add_argument("--voxel-size", type=int, default=None)
    ap.add_argument("--num-views", type=int, default=None)
    ap.add_argument("--image-size", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-voxel-config5", action="store_true", help="skip the child leg that times the 3D-conv forwards at 64^3 x 64")
    ap.add_argument("--cpu-batch", type=int, default=0, help="batch of the CPU leg (0: the per-GPU batch, halved until 2 + 5 steps fit the budget)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU leg (0: the fastest of 16 / 32 / 64 / physical cores / all CPUs, probed)")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    ap.add_argument("--resident-batches", type=int, default=8)
    ap.add_argument("--preroll", type=int, default=40)
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each; value = the median window")
    a = ap.parse_args(argv)
    text, image, voxel, V, nv, S, pb = CONFIGS[a.config]
    a.text, a.image, a.voxel = text, image, voxel
    a.voxel_size = a.voxel_size or V
    a.num_views = a.num_views or nv
    a.image_size = a.image_size or S
    a.per_gpu_batch = a.per_gpu_batch or pb
    return a


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N rank processes from here, as
    CHILDREN of this process, which has not touched the GPU (a process that has must never exec another program on this pool).  Rank
    0's stdout is passed through (its last line is the JSON line), the exit code is the worst of the ranks'; when one rank dies the
    others - stuck in a rendezvous or a collective by then - are ended by PID."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p_ in list(live):
            code = p_.poll()
            if code is None:
                continue
            live.remove(p_)
            if code != 0:
                rc = rc or code
                for q in live:                       # our own children, by PID
                    q.terminate()
    return rc


def main():
    a = parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))                     # before anything in this process initialises the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    # TRICOLO_DP_SHARE_GPU=1: every rank on cuda:0, collectives over gloo through host memory (parallel.py) - runs the N > 1 code path
    # (rank-offset row slices, three graphs around the collectives) on a one-GPU box; a validation aid, never a measurement
    share = os.environ.get("TRICOLO_DP_SHARE_GPU", "0") == "1"
    ndev = torch.cuda.device_count()                 # (counting devices does not initialise the GPU)
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if share:
        local_rank = 0
    if local_rank >= ndev:
        raise SystemExit(f"bench.py --gpus {a.gpus}: rank {rank} needs device {local_rank} but this box has {ndev} GPU(s) "
                         f"(need {a.gpus} devices; TRICOLO_DP_SHARE_GPU=1 runs all ranks on one GPU over gloo for validation)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("TRICOLO_FORCE_DIST", "0") == "1"      # world-of-one run of the data-parallel code path
    if world > 1 or force_dist:
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    if a.modes is None:
        extra = [m for m in ("bf16x3", "bf16", "f16") if m != a.precision] if world == 1 else []
    else:
        extra = [m for m in a.modes.split(",") if m and m != a.precision]

    head, vox_roof, cfg = run_mode(a, a.precision, world, rank, device, want_voxel_roofline=True)
    modes = {}
    for m in extra:
        if world == 1 and not dist.is_initialized():
            # each extra mode in a CHILD process of its own: a second mode timed in this process inherits the allocator / physical
            # memory state the first one left (the fp32-storage parity mode ran 6.8-8.3 ms after an f16 leg, 6.8 ms on its own)
            try:
                modes[m] = run_mode_in_child(a, m)
            except Exception as e:                                 # noqa: BLE001  a box that cannot spawn: time the mode here instead
                print(f"[bench] child process for mode {m} failed ({e}); timing it in this process", file=sys.stderr)
                modes[m], _, _ = run_mode(a, m, world, rank, device, want_voxel_roofline=False)
                modes[m]["process"] = "same (child failed)"
        else:
            modes[m], _, _ = run_mode(a, m, world, rank, device, want_voxel_roofline=False)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:             # the CPU leg is an N = 1 artefact (task contract)
        cpu = cpu_baseline(a, cfg)

    vox5 = None
    if rank == 0 and world == 1 and a.config == 4 and not a.no_voxel_config5 and not a.no_cpu_baseline:
        try:                                                            # (default line only: A/B runs pass --no-cpu-baseline)
            vox5 = voxel_fwd_config5_in_child()
        except Exception as e:                                         # noqa: BLE001  reported, never fatal for the headline
            vox5 = {"error": str(e)[-500:]}

    if rank == 0:
        gb = a.per_gpu_batch * world
        out = {
            "metric": "trimodal training samples/sec (32^3 voxel + 6-view)" if a.config == 4 else f"training samples/sec, BASELINE config {a.config}",
            "value": head["value"], "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "ms_per_step_windows": head["ms_per_step_windows"], "step_tflops": head["step_tflops"],
            "step_tflops_note": head["step_tflops_note"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"], "data": "synthetic",
            "config": {"workload": workload_name(a), "global_batch": gb, "per_gpu_batch": a.per_gpu_batch, "parallelism": f"dp{world}",
                       "resident_batches": a.resident_batches,
                       "switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("TRICOLO_")},   # every kernel / tile-rule switch that was set
                       "precision_mode": a.precision, "hip_graph": head["hip_graph"], "final_loss": head["final_loss"],
                       "roofline_3dconv_fwd_primary": "frac_active_rows_raw_events (FLOPs of the active rows over the raw single-launch event times; "
                                                      "`frac` = executed FLOPs on the same times)",
                       "parity": "f16 mode: step-0 losses and embeddings within 1e-3 of the fp32 reference on BASELINE configs 1,3,4,5 "
                                 "(tests/test_gpu_modules.py::test_f16_mode_meets_the_1e3_parity_bound); bf16x3: ~1e-5; bf16: outside 1e-3"},
            "roofline": head["roofline"], "modes": modes, "roofline_3dconv_fwd": vox_roof, "roofline_3dconv_fwd_config5": vox5, "cpu_baseline": cpu,
            "data_parallel": head.get("data_parallel"), "nonfinite_guard": head.get("nonfinite_guard"),
        }
        for m, d_ in modes.items():                                   # flat copies: the driver keeps top-level keys
            out[f"value_{m}"] = d_["value"]
            out[f"ms_per_step_{m}"] = d_["ms_per_step"]
        if vox_roof:
            vox_roof["primary"] = "frac_active_rows_raw_events"
            out["frac_3dconv_fwd"] = vox_roof["frac_active_rows_raw_events"]
        if vox5 and "error" not in vox5:
            out["frac_3dconv_fwd_config5"] = vox5["frac_active_rows_raw_events"]
            out["frac_3dconv_fwd_config5_executed"] = vox5["frac"]
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
