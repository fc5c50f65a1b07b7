"""GPU parity of the HIP-backed encoder / loss / container modules against the golden vectors produced by the REAL
reference classes (tests/golden, oracle/make_golden.py) with identical recipe weights and identical synthetic inputs.

Tolerances (fp32 reference vs bf16x3 split-MFMA path): embeddings 2e-4 abs on unit-norm rows, losses 1e-3 (the
north-star bound), gradient norms 2e-3 relative for the voxel / text towers.  Sampled gradient entries: within 5 x rtol of the tensor's rms (voxel / text towers:
1e-2 rms; measured <= 7.5e-3), 15 x rtol = 0.3 rms for the ResNet tower (measured 0.23).  The ResNet-18 tower's parameter gradients
are ill-conditioned through ReLU / max-pool / view-max routing: perturbing the weights of the fp32 CPU oracle itself by
1.5e-5 relative (the split-bf16 operand error) moves its gradient norms by up to 0.7 % (measured in the build
container), so that tower's bound is 2e-2 on norms; every kernel's backward is separately pinned EXACTLY on integer
data in test_gpu_ops.py.  The plain-bf16 mode is checked separately with its own stated bound."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tricolo_amd import config as tcfg, ops
    from tricolo_amd.loss.nt_xent import NTXentLoss
    from tricolo_amd.model.module.img_encoder.mv_cnn import MVCNNEncoder
    from tricolo_amd.model.module.text_encoder.bigru import BiGRUEncoder
    from tricolo_amd.model.module.text_encoder.clip_text import CLIPTextEncoder
    from tricolo_amd.model.module.voxel_encoder.sparse_cnn import SparseCNNEncoder
    from tricolo_amd.model.tricolo_net import TriCoLoNet

from oracle.recipe import fill_module, probe
from tricolo_amd.data import synthetic as syn

DEV = "cuda"
EMB_TOL, LOSS_TOL, GRAD_RTOL = 2e-4, 1e-3, 2e-3


def _sha(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    return h.hexdigest()


import json
import os

_REPORT = {}


def _report(key, value):
    """Measured deviations are collected in gpurun_out/parity_report.json (copied to profiles/ for the record)."""
    _REPORT[key] = value
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_report.json", "w") as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _check_grads(module, g, prefix, rtol=GRAD_RTOL, sample_mult=5.0, tag=None):
    """Gradient norms within rtol and the 16 sampled entries of every parameter gradient within sample_mult * rtol * rms."""
    bad, worst_n, worst_s = [], 0.0, 0.0
    for name, p in module.named_parameters():
        key = f"{prefix}gradnorm/{name}"
        if key not in g:
            continue
        assert p.grad is not None, name
        n, s = probe(p.grad.cpu())
        ref_n, ref_s = float(g[key]), g[f"{prefix}gradsample/{name}"]
        scale = max(ref_n / np.sqrt(p.numel()), 1e-8)
        dn, ds = abs(n - ref_n) / max(ref_n, 1e-6), float(np.abs(s - ref_s).max()) / scale
        worst_n, worst_s = max(worst_n, dn), max(worst_s, ds)
        if dn > rtol or ds > sample_mult * rtol + 1e-7 / scale:
            bad.append((name, n, ref_n, dn, ds))
    if tag:
        _report(f"grads/{tag}", {"worst_rel_norm_diff": worst_n, "worst_sample_diff_over_rms": worst_s, "rtol": rtol,
                                 "sample_bound_over_rms": sample_mult * rtol})
    assert not bad, bad


@pytest.fixture(autouse=True)
def _precision():
    ops.set_default_precision("bf16x3")
    yield
    ops.set_default_precision("bf16x3")


def test_bigru_matches_reference(golden):
    g = golden("bigru")
    m = BiGRUEncoder(syn.DEFAULT_VOCAB, 512)
    fill_module(m, prefix="text_encoder.")
    m = m.to(DEV)
    batch = syn.make_batch(8, voxel_size=None, num_views=None, seed=syn.BASE_SEED + 6)
    z = m(batch["tokens"].to(DEV), batch)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["z"], atol=EMB_TOL)
    (z * torch.from_numpy(g["upstream"]).to(DEV)).sum().backward()
    _check_grads(m, g, "", tag="bigru")
    zp = m(torch.zeros((2, 96), dtype=torch.int32, device=DEV), {})
    np.testing.assert_allclose(zp.detach().cpu().numpy(), g["z_allpad"], atol=EMB_TOL)


def test_clip_text_matches_reference(golden):
    g = golden("clip_text")
    m = CLIPTextEncoder(out_dim=512)
    fill_module(m, prefix="text_encoder.")
    m = m.to(DEV).eval()
    batch = syn.batch_to_device(syn.make_batch(8, voxel_size=None, num_views=None, clip_text=True, seed=syn.BASE_SEED + 5), DEV)
    z = m(batch["tokens"], batch)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["z"], atol=2e-4)
    with pytest.raises(UnboundLocalError):
        m(batch["tokens"], {})


def test_clip_text_train_mode_dropout_with_an_injected_mask(monkeypatch):
    """Train mode: Linear -> ReLU -> Dropout(0.1) -> Linear (clip_text.py:9-14).  The dropout mask of the GPU generator cannot be
    reproduced on the CPU, so the mask is INJECTED: F.dropout in the module's namespace is replaced by a fixed Bernoulli(0.9) mask
    scaled by 1 / 0.9 (exactly what nn.Dropout computes), and forward + every parameter gradient are compared with a float64
    restatement of the same four lines using that mask."""
    import tricolo_amd.model.module.text_encoder.clip_text as ct
    gen = torch.Generator().manual_seed(41)
    m = CLIPTextEncoder(out_dim=512).to(DEV).train()
    B = 8
    x = torch.randn(B, 768, generator=gen)
    x = x / x.norm(dim=1, keepdim=True)                             # unit-norm CLIP vectors (extract_clip_feats.py:30-31)
    mask = (torch.rand(B, 512, generator=gen) < 0.9).float()
    calls = []

    def fake_dropout(h, p, training):
        calls.append((p, training))
        return h * (mask.to(h.device) / (1.0 - p))
    monkeypatch.setattr(ct.F, "dropout", fake_dropout)
    z = m(torch.zeros((B, 96), dtype=torch.int32, device=DEV), {"clip_embeddings_text": x.to(DEV)})
    assert calls == [(0.1, True)]
    up = torch.randn(B, 512, generator=gen)
    (z * up.to(DEV)).sum().backward()
    w0, b0, w1, b1 = (t.detach().cpu().double().requires_grad_() for t in (m.mlp[0].weight, m.mlp[0].bias, m.mlp[3].weight, m.mlp[3].bias))
    h = torch.relu(x.double() @ w0.t() + b0) * (mask.double() / 0.9)
    zr = h @ w1.t() + b1
    (zr * up.double()).sum().backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), zr.detach().numpy(), atol=2e-5)
    for p_, r_ in ((m.mlp[0].weight, w0), (m.mlp[0].bias, b0), (m.mlp[3].weight, w1), (m.mlp[3].bias, b1)):
        np.testing.assert_allclose(p_.grad.cpu().numpy(), r_.grad.numpy(), atol=3e-5, rtol=2e-4)
    # and the real thing: nn.Dropout semantics hold statistically (10 % of the hidden units zeroed, the rest scaled by 1 / 0.9)
    monkeypatch.undo()
    with torch.no_grad():
        big = torch.randn(64, 768, device=DEV)
        hid = torch.relu(big @ m.mlp[0].weight.t() + m.mlp[0].bias)
        out = ct.F.dropout(hid, 0.1, True)
        alive = hid > 0                                             # ReLU zeros say nothing about the mask
        assert 0.87 < (out[alive] != 0).float().mean().item() < 0.93
        np.testing.assert_allclose(out[out != 0].cpu().numpy(), (hid[out != 0] / 0.9).cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("tag,V,B,seed", [("v32", 32, 8, 1), ("v64", 64, 2, 8)])
def test_voxel_encoder_matches_reference(golden, tag, V, B, seed):
    g = golden("voxel")
    m = SparseCNNEncoder(V, 32, 512, 512)
    fill_module(m, prefix="voxel_encoder.")
    m = m.to(DEV)
    batch = syn.batch_to_device(syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + seed), DEV)
    assert _sha(batch["voxels"]["locs"], batch["voxels"]["feats"]) == str(g[f"{tag}/input_sha"])
    z = m(batch["voxels"], B)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g[f"{tag}/z"], atol=EMB_TOL)
    (z * torch.from_numpy(g[f"{tag}/upstream"]).to(DEV)).sum().backward()
    _check_grads(m, g, f"{tag}/", tag=f"voxel_{tag}")
    # running statistics after one train-mode forward
    for name, v in m.state_dict().items():
        if "running" in name:
            n, s = probe(v.cpu())
            np.testing.assert_allclose(s, g[f"{tag}/after/wsample/{name}"], rtol=1e-4, atol=1e-5, err_msg=name)


def test_voxel_encoder_empty_sample_and_eval_mode(golden):
    g = golden("voxel")
    m = SparseCNNEncoder(32, 32, 512, 512)
    fill_module(m, prefix="voxel_encoder.")
    m = m.to(DEV)
    batch = syn.make_batch(3, voxel_size=32, num_views=None, seed=syn.BASE_SEED + 9)
    keep = batch["voxels"]["locs"][:, 0] != 1
    vox = {"locs": batch["voxels"]["locs"][keep].to(DEV), "feats": batch["voxels"]["feats"][keep].to(DEV)}
    with torch.no_grad():
        z = m(vox, 3)
    np.testing.assert_allclose(z.cpu().numpy(), g["empty1/z"], atol=EMB_TOL)
    # eval mode: running statistics, no autograd node; compare with the oracle restatement in eval mode
    from oracle.modules import SparseCNNRef
    ref = SparseCNNRef(32, 32, 512, 512)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    ref.eval(); m.eval()
    zr = ref({"locs": vox["locs"].cpu(), "feats": vox["feats"].cpu()}, 3)
    with torch.no_grad():
        ze = m(vox, 3)
    np.testing.assert_allclose(ze.cpu().numpy(), zr.detach().numpy(), atol=EMB_TOL)


@pytest.mark.parametrize("tag,B,nv,S", [("v6s128", 8, 6, 128), ("v12s224", 2, 12, 224)])
def test_mvcnn_encoder_matches_reference(golden, tag, B, nv, S):
    g = golden("mvcnn")
    m = MVCNNEncoder(512, 512, "resnet18", nv)
    assert len(m.state_dict()) == 126
    fill_module(m, prefix="image_encoder.")
    m = m.to(DEV)
    batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
    assert _sha(batch["images"]) == str(g[f"{tag}/input_sha"])
    z = m(batch["images"].flatten(end_dim=1).to(DEV), batch)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g[f"{tag}/z"], atol=EMB_TOL)
    (z * torch.from_numpy(g[f"{tag}/upstream"]).to(DEV)).sum().backward()
    # measured (profiles/r2/parity_report.json): norms within 8e-3, sampled entries within 0.23 rms - single entries move with the
    # ReLU / max routing flips described in the module docstring; the kernels' lo terms are pinned by test_gpu_ops.py
    _check_grads(m, g, f"{tag}/", rtol=2e-2, sample_mult=15.0, tag=f"mvcnn_{tag}")
    for name, v in m.state_dict().items():
        if "running" in name:
            n, s = probe(v.cpu())
            np.testing.assert_allclose(s, g[f"{tag}/after/wsample/{name}"], rtol=1e-4, atol=1e-5, err_msg=name)


def test_mvcnn_gradients_within_measured_conditioning(golden):
    """The ResNet-18 tower's parameter gradients against FLOAT64 gradients of the reference wrapper, each tensor bounded by its own
    measured conditioning (tests/golden/mvcnn_sens.npz, oracle/make_mvcnn_sensitivity.py: how far the float64 gradient itself moves
    under a 1.5e-5 relative weight perturbation, the size of the split-bf16 operand error).  Round-3 finding: that perturbation moves
    the float64 gradients by 4.4 % in L2 (median over tensors; ReLU masks of near-zero activations flip) while norms move 0.2 % - the
    HIP path sits at the same 4 % (its forward is within 6e-5), so the element-wise pin of THIS fixture cannot be tighter than that;
    the tight pin of the backward composition is test_mvcnn_backward_replay_with_forced_routing below.  Checked here: embeddings,
    norms and probes within 3 x their sensitivity, and the 40 BatchNorm weight / bias gradients WHOLE within 3 x their L2
    sensitivity (well-conditioned tensors such as layer4.1's biases are thereby pinned to ~1e-3)."""
    g = golden("mvcnn_sens")
    tag, B, nv, S = "v6s128", 8, 6, 128
    m = MVCNNEncoder(512, 512, "resnet18", nv)
    fill_module(m, prefix="image_encoder.")
    m = m.to(DEV)
    batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
    assert _sha(batch["images"]) == str(g[f"{tag}/input_sha"])
    z = m(batch["images"].flatten(end_dim=1).to(DEV), batch)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g[f"{tag}/z64"], atol=EMB_TOL)
    up = torch.randn((B, 512), generator=torch.Generator().manual_seed(13))
    (z * up.to(DEV)).sum().backward()
    bad, worst, worst_l2, nvec = [], 0.0, 0.0, 0
    for name, p in m.named_parameters():
        ref_n, ref_s = float(g[f"{tag}/gradnorm64/{name}"]), g[f"{tag}/gradsample64/{name}"]
        sn, ss, sl = float(g[f"{tag}/sens_norm/{name}"]), float(g[f"{tag}/sens_sample/{name}"]), float(g[f"{tag}/sens_l2/{name}"])
        bn, bs, bl = 3.0 * max(sn, 0.25 * sl) + 2e-4, 3.0 * ss + 2e-3, 3.0 * sl + 2e-4
        n, smp = probe(p.grad.cpu())
        rms = max(ref_n / np.sqrt(p.numel()), 1e-30)
        dn, ds = abs(n - ref_n) / max(ref_n, 1e-30), float(np.abs(smp.astype(np.float64) - ref_s).max()) / rms
        worst = max(worst, dn / bn, ds / bs)
        if dn > bn or ds > bs:
            bad.append((name, "norm/probe", dn, bn, ds, bs))
        key = f"{tag}/grad64/{name}"
        if key in g:                                                # whole vector stored: element-wise L2
            ref_v = g[key]
            dl = float(np.linalg.norm(p.grad.detach().double().cpu().numpy().reshape(-1) - ref_v) / max(np.linalg.norm(ref_v), 1e-300))
            worst_l2 = max(worst_l2, dl / bl)
            nvec += 1
            if dl > bl:
                bad.append((name, "L2", dl, bl))
    _report("grads/mvcnn_conditioning", {"worst_ratio_to_bound_norm_probe": worst, "worst_ratio_to_bound_l2": worst_l2, "vectors_checked_whole": nvec})
    assert nvec >= 40 and not bad, bad


# Per-tensor relative L2 bounds of the forced-routing replays below, by precision mode.  bf16x3 (split operands, fp32 storage): measured
# 6.4e-5 on the image tower - the composition of the backward is pinned there.  The 16-bit modes run the SAME kernels (templates on the
# storage type) and add the storage rounding of every activation and gradient tensor, eps = 2^-11 (f16) / 2^-8 (bf16) relative per
# element; a parameter gradient is a sum over positions of products of such tensors, so its error is ~eps times a cancellation factor
# (BatchNorm gammas / betas of the deep layers sum ~3 k terms of both signs: measured 10 eps on net_1.7.1.bn1.weight, 1-4 eps on the conv
# weights, 3.4 eps on the stem conv).  Bound: 12 eps - a 1 % error in any single tensor (the size of bug the flat norm / probe bounds of
# test_16bit_modes_backward_on_bi_v cannot see) is twice the f16 bound.
REPLAY_BOUND = {"bf16x3": 5e-4, "f16": 12 * 2.0 ** -11, "bf16": 12 * 2.0 ** -8}


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "bf16"])
def test_mvcnn_backward_replay_with_forced_routing(prec):
    """A tight pin of the image tower's BACKWARD composition in EVERY precision mode (the flat gradient bounds could not see
    a 1 % bug; VERDICT r3: the 16-bit backward kernels - halo, c64, s2d, DMA weight gradients, the fused stem - were pinned per kernel on
    integer data only, never as a composed tower).  The ResNet's gradient is discontinuous in its activations - a ReLU mask, a max-pool
    tap or a view arg-max that flips under a 1e-5 forward difference changes gradient elements by O(1).  Here the float64 oracle is
    replayed with the ROUTING FORCED to the HIP forward's own (of the mode under test): every block ReLU multiplies by the mask of the
    HIP path's stored activations, the stem's ReLU + 3x3/2 max-pool gathers the HIP path's winning tap (its stored arg-max map) times the
    mask of its pooled output, the view max gathers the HIP path's arg-max view, the head's ReLU uses the HIP path's hidden-layer mask.
    What is left is a smooth function of the weights, and
    the HIP gradients of EVERY tensor - stem conv1 / bn1 included - must agree with it element-wise: relative L2 per tensor <=
    REPLAY_BOUND[mode]."""
    from oracle import modules as om
    ops.set_default_precision(prec)
    B, nv, S = 8, 6, 128
    batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
    up = torch.randn((B, 512), generator=torch.Generator().manual_seed(13))
    m = MVCNNEncoder(512, 512, "resnet18", nv)
    fill_module(m, prefix="image_encoder.")
    m = m.to(DEV)
    images = batch["images"].flatten(end_dim=1)
    z = m(images.to(DEV), batch)
    (z * up.to(DEV)).sum().backward()
    with torch.no_grad():                                            # the same forward again, keeping its activations
        _, saved = m._forward_impl(images.to(DEV), save=True)
    blocks = saved["lower"]["blocks"] + saved["upper"]["blocks"]
    masks = []
    for sv in blocks:                                                # (x, y1, co1, g1, a1, y2, co2, g2, yd, cod, gd, out), channels-last
        for tns in (sv[4], sv[-1]):
            masks.append((tns[:, 0] > 0).permute(0, 3, 1, 2).cpu())  # -> [N, C, H, W] bool
    arg = saved["upper"]["arg"].cpu().long()                         # [B, 512] winning view
    parg = saved["lower"]["stem"][4][:, 0].permute(0, 3, 1, 2).cpu().long()          # [N, C, Ho, Wo] winning tap kh * 3 + kw of each window
    pmask = (blocks[0][0][:, 0] > 0).permute(0, 3, 1, 2).cpu()                       # pooled output > 0  <=>  the winner passed the ReLU

    class Forced(torch.nn.Module):
        def __init__(self, it):
            super().__init__()
            self.it = it

        def forward(self, x):
            return x * next(self.it).to(x.dtype)

    class ForcedStemPool(torch.nn.Module):
        """relu -> MaxPool2d(3, 2, 1) with the winner given: out[n, c, oh, ow] = t[n, c, 2 oh - 1 + kh, 2 ow - 1 + kw] * (winner > 0)."""

        def forward(self, t):
            N, C, H, W = t.shape
            Ho, Wo = parg.shape[-2:]
            ih = (2 * torch.arange(Ho).view(1, 1, Ho, 1) - 1 + parg // 3).clamp(0, H - 1)
            iw = (2 * torch.arange(Wo).view(1, 1, 1, Wo) - 1 + parg % 3).clamp(0, W - 1)
            g_ = torch.gather(t.reshape(N, C, H * W), 2, (ih * W + iw).reshape(N, C, Ho * Wo)).view(N, C, Ho, Wo)
            return g_ * pmask.to(t.dtype)

    ref = om.MVCNNRef(512, 512, "resnet18", nv)
    fill_module(ref, prefix="image_encoder.")
    ref = ref.double()
    it = iter(masks)
    ref.net_1[2] = torch.nn.Identity()                               # stem ReLU + max-pool: forced together
    ref.net_1[3] = ForcedStemPool()
    for li in (4, 5, 6, 7):
        for blk in ref.net_1[li]:
            blk.relu = Forced(it)                                    # called twice per block: after bn1, after the residual sum
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    y = ref.net_1(images.double())                                   # [N, 512, 1, 1]
    y = y.view(B, nv, 512)
    y = torch.gather(y, 1, arg.view(B, 1, 512)).view(B, 512)        # the HIP path's view instead of torch.max
    hmask = (saved["upper"]["h"] > 0).double().cpu()                 # the head's ReLU (mlp[1]) is routing too
    zr = torch.nn.functional.normalize(ref.mlp[2](ref.mlp[0](ref.net_2(y)) * hmask), dim=1)
    assert next(it, None) is None
    np.testing.assert_allclose(z.detach().cpu().numpy(), zr.detach().numpy(), atol=EMB_TOL if prec == "bf16x3" else 5e-3)
    (zr * up.double()).sum().backward()
    rg = dict(ref.named_parameters())
    worst, worst_name, bad, table = 0.0, "", [], {}
    for name, p in m.named_parameters():
        a = rg[name].grad
        dl = float((p.grad.detach().double().cpu() - a).norm() / a.norm().clamp_min(1e-300))
        table[name] = dl
        if dl > worst:
            worst, worst_name = dl, name
        if dl > REPLAY_BOUND[prec]:
            bad.append((name, dl))
    conv = sorted(v for k, v in table.items() if k.endswith("weight") and v == v and ("conv" in k or "downsample.0" in k or k == "net_1.0.weight"))
    _report(f"grads/mvcnn_forced_routing_replay/{prec}", {"worst_rel_l2": worst, "worst_tensor": worst_name, "stem_conv1_rel_l2": table["net_1.0.weight"],
                                                          "conv_weights_median_rel_l2": conv[len(conv) // 2], "conv_weights_max_rel_l2": conv[-1],
                                                          "bound": REPLAY_BOUND[prec]})
    assert not bad, bad


@pytest.mark.parametrize("prec", ["bf16x3", "f16", "bf16"])
def test_voxel_backward_replay_with_forced_routing(prec):
    """The voxel tower's twin of the test above: float64 oracle (oracle/spconv_dense.py semantics) replayed with the
    ReLU masks and the 2^3 max-pool winners FORCED to the HIP forward's - recomputed here from the tensors the HIP backward itself routes
    by (conv output y, BatchNorm coefficients, site mask, pooled maximum: first child in (d, h, w) scan order whose rounded post-ReLU
    value equals the pooled maximum and is > 0, bn_pool.hip pool3d_bwd_route_kernel).  Every parameter gradient of the tower must then
    agree element-wise: relative L2 per tensor <= REPLAY_BOUND[mode].  Covers the brick kernels (levels 0 / 1), the row-list forward /
    data / weight gradients (levels 1-4) and the fused pool-routing + BatchNorm-backward passes in the 16-bit modes."""
    from tests.replay import voxel_forced_replay
    ops.set_default_precision(prec)
    table, unmatched, zdiff = voxel_forced_replay(32, 8)
    assert zdiff <= (EMB_TOL if prec == "bf16x3" else 5e-3)
    worst_name = max(table, key=table.get)
    worst = table[worst_name]
    bad = [(n, v) for n, v in table.items() if not v <= REPLAY_BOUND[prec]]
    _report(f"grads/voxel_forced_routing_replay/{prec}", {"worst_rel_l2": worst, "worst_tensor": worst_name, "winners_taken_by_argmax": unmatched,
                                                          "bound": REPLAY_BOUND[prec]})
    assert not bad, bad


@pytest.mark.parametrize("prec", ["bf16x3", "f16"])
def test_voxel_masked_tile_path_is_pinned_like_the_compact_row_path(monkeypatch, prec):
    """TRICOLO_VOXEL_COMPACT=0 (the A/B partner: site masks + tile skipping instead of compact active-row lists) through the same
    forced-routing float64 replay as the default path, with NaNs in the allocator's free blocks (with fp32 storage the
    level-0 weight gradient contracted over dy rows the BatchNorm-backward apply pass had been told to leave unwritten - garbage or NaN
    gradients for sparseModel['0'].weight).  The two paths are not compared with each other: a last-bit forward difference flips ReLU /
    max-pool routing and moves single gradient elements by O(1) (measured 1.2 % between the paths in f16); each must match ITS OWN routing."""
    from tests.replay import voxel_forced_replay
    ops.set_default_precision(prec)
    monkeypatch.setenv("TRICOLO_VOXEL_COMPACT", "0")
    table, unmatched, zdiff = voxel_forced_replay(32, 8, poison=True)
    assert zdiff <= (EMB_TOL if prec == "bf16x3" else 5e-3)
    worst_name = max(table, key=lambda k: table[k] if table[k] == table[k] else float("inf"))
    _report(f"grads/voxel_masked_tile_path_replay/{prec}", {"worst_rel_l2": table[worst_name], "worst_tensor": worst_name, "bound": REPLAY_BOUND[prec]})
    bad = [(n, v) for n, v in table.items() if not v <= REPLAY_BOUND[prec]]
    assert not bad, bad


def test_ntxent_module_autograd(golden):
    g = golden("ntxent")
    za = torch.from_numpy(g["b8/za"]).to(DEV).requires_grad_()
    zb = torch.from_numpy(g["b8/zb"]).to(DEV).requires_grad_()
    loss = NTXentLoss(0.1, 0.25)(za, zb)
    (2.0 * loss).backward()
    assert abs(loss.item() - float(g["b8/loss"])) < 2e-5
    np.testing.assert_allclose(za.grad.cpu().numpy(), 2 * g["b8/dza"], atol=4e-6)
    np.testing.assert_allclose(zb.grad.cpu().numpy(), 2 * g["b8/dzb"], atol=4e-6)


STEP_CASES = [
    ("cfg1_biV", "BiGRUEncoder", None, "SparseCNNEncoder", 32, None, 128, 8, 1, False),
    ("cfg3_biI", "BiGRUEncoder", "MVCNNEncoder", None, 32, 6, 128, 8, 3, False),
    ("cfg4_tri", "BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128, 8, 4, False),
    ("cfg5_tri64", "CLIPTextEncoder", "MVCNNEncoder", "SparseCNNEncoder", 64, 12, 224, 2, 5, True),
]


def _build_net(text, image, voxel, V, nv, S, optimizer="tricolo_amd.optim.FusedAdam"):
    ov = [f"model.text_encoder={text}", f"model.image_encoder={image or 'null'}", f"model.voxel_encoder={voxel or 'null'}",
          "data=synthetic", f"data.voxel_size={V}", f"data.num_views={nv}", f"data.image_size={S}",
          f"optimizer._target_={optimizer}", "experiment_name=test"]
    cfg = tcfg.compose(overrides=ov)
    net = TriCoLoNet(cfg)
    fill_module(net)
    if text == "CLIPTextEncoder":
        net.text_encoder.mlp[2].eval()
    return net.to(DEV), cfg


@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_training_steps_match_reference(golden, case):
    tag, text, image, voxel, V, nv, S, B, seed_off, clip_text = case
    g = golden(f"step_{tag}")
    net, cfg = _build_net(text, image, voxel, V, nv or 6, S)
    batch = syn.make_batch(B, voxel_size=V if voxel else None, num_views=nv if image else None, image_size=S,
                           clip_text=clip_text, seed=syn.BASE_SEED + seed_off)
    batch = syn.batch_to_device(batch, DEV)
    opt = net.configure_optimizers()
    report = {}
    for step in range(4):
        opt.zero_grad(set_to_none=True)
        emb = net(batch)
        for k in emb:
            emb[k].retain_grad()
        losses = net._calculate_losses(emb, "train_loss")
        total = losses["train_loss/total_loss"]
        report[step] = (total.item(), float(g[f"step{step}/total_loss"]))
        if step == 0:
            for k, v in losses.items():
                assert abs(v.item() - float(g[f"step0/{k}"])) < LOSS_TOL, (k, v.item(), float(g[f"step0/{k}"]))
            for k, v in emb.items():
                np.testing.assert_allclose(v.detach().cpu().numpy(), g[f"emb/{k}"], atol=EMB_TOL, err_msg=k)
        if step == 3:
            break
        total.backward()
        if step == 0:
            for k, v in emb.items():
                np.testing.assert_allclose(v.grad.cpu().numpy(), g[f"demb/{k}"], atol=2e-5, err_msg=k)
            _check_grads(net, g, "", rtol=2e-2, sample_mult=15.0 if image else 5.0, tag=f"step_{tag}")
        opt.step()
    print(tag, report)
    _report(f"trajectory/{tag}", {str(k): {"hip": v[0], "reference": v[1], "abs_diff": abs(v[0] - v[1])} for k, v in report.items()})
    # after Adam updates (sign-like first steps amplify tiny gradient differences) the bound is looser and stated:
    for step in (1, 2, 3):                                   # measured <= 7.7e-3 (profiles/r2/parity_report.json)
        assert abs(report[step][0] - report[step][1]) < 1e-2, report


def test_fused_adam_equals_torch_adam_on_the_same_net(golden):
    g = golden("step_cfg1_biV")
    net, cfg = _build_net("BiGRUEncoder", None, "SparseCNNEncoder", 32, 6, 128, optimizer="torch.optim.Adam")
    batch = syn.batch_to_device(syn.make_batch(8, voxel_size=32, num_views=None, seed=syn.BASE_SEED + 1), DEV)
    opt = net.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam)
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        loss = net.training_step(batch, 0)
        assert abs(loss.item() - float(g[f"step{step}/total_loss"])) < (LOSS_TOL if step == 0 else 2e-2)
        loss.backward()
        opt.step()


def test_graph_split_dp_step_equals_eager_dp_step(monkeypatch):
    """parallel.GraphedDPStep (three HIP graphs around the two eager collectives, the N > 1 bench path) against the
    eager dp_training_step from identical weights on the same batch, in a one-rank RCCL world with the data-parallel
    code path forced on.  Same kernels, same order: the loss trajectories must agree to fp32 round-off."""
    import torch.distributed as dist
    from tricolo_amd import parallel
    monkeypatch.setenv("TRICOLO_FORCE_DIST", "1")
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1").items():
        monkeypatch.setenv(k, v)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        net_a, _ = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 2, 64)
        net_b, _ = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 2, 64)     # same recipe weights
        batch = syn.batch_to_device(syn.make_batch(8, voxel_size=32, num_views=2, image_size=64, seed=syn.BASE_SEED + 31), DEV)
        opt_a, opt_b = net_a.configure_optimizers(), net_b.configure_optimizers()
        for o in (opt_a, opt_b):
            o.prepare()
        eager = []
        for _ in range(5):
            eager.append(parallel.dp_training_step(net_a, batch, opt_a)["train_loss/total_loss"].item())
        # the graph-split step needs the same warm-up the bench does (lazy buffers, RCCL communicator) - done on net_b too
        warm = [parallel.dp_training_step(net_b, batch, opt_b)["train_loss/total_loss"].item() for _ in range(2)]
        torch.cuda.synchronize()
        gstep = parallel.GraphedDPStep(net_b, opt_b, batch)
        graphed = warm + [gstep.replay().item() for _ in range(3)]
        np.testing.assert_allclose(graphed, eager, rtol=2e-4, atol=2e-4)
        assert eager[-1] < eager[0]
        # the overlapped variant: image tower as two autograd nodes, backward in two stages around the exposed feature map, the
        # early gradient ranges reduced asynchronously under the second stage - eagerly and as four graphs
        net_c, _ = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 2, 64)
        opt_c = net_c.configure_optimizers()
        opt_c.prepare()
        split = parallel.BackwardSplit.for_net(net_c)
        assert split is not None and len(split.late_params) == 30 + 19        # image stem + layer1-2, and the voxel tower
        n_late = sum(p.numel() for p in split.late_params)
        n_all = sum(p.numel() for p in net_c.parameters())
        assert n_late < 0.35 * n_all                                         # two thirds of the gradient bytes reduce early
        over = [parallel.dp_training_step(net_c, batch, opt_c, split=split)["train_loss/total_loss"].item() for _ in range(2)]
        torch.cuda.synchronize()
        gover = parallel.GraphedDPStep(net_c, opt_c, batch, split=split)
        over += [gover.replay().item() for _ in range(3)]
        np.testing.assert_allclose(over, eager, rtol=2e-4, atol=2e-4)
    finally:
        dist.destroy_process_group()


def test_retrieval_metrics_on_device_match_reference(golden):
    """compute_metrics with the device ranking (tri_retrieval_topk) against the numbers and the top-5 index
    matrix the REAL reference compute_metrics produced for the same embeddings (tests/golden/retrieval.npz), and against
    the host algorithm on a set with exact ties (duplicate shape rows)."""
    from tricolo_amd.evaluation.eval_retrieval import compute_metrics
    g = golden("retrieval")
    shape = g["text"] * 0 + g["image"] + g["voxel"]
    tuples = [(None, "cat", str(mid), g["text"][i], shape[i]) for i, mid in enumerate(g["model_ids"])]
    r = compute_metrics("Text2Shape", {"caption_embedding_tuples": tuples})
    np.testing.assert_array_equal(r["indices"], g["indices"])
    np.testing.assert_allclose(r["recall_rate"], g["recall_rate"], atol=0)
    np.testing.assert_allclose(r["ndcg"], g["ndcg"], atol=1e-12)
    np.testing.assert_allclose(r["precision"], g["precision"], atol=1e-12)
    assert abs(r["mrr"] - float(g["mrr"])) < 1e-12
    # exact ties: every shape row appears twice under two ids; descending order with the higher index first
    rng = np.random.default_rng(3)
    base = rng.standard_normal((40, 64)).astype(np.float32)
    shp = np.concatenate([base, base])
    txt = base[rng.integers(0, 40, 96)] + 0.01 * rng.standard_normal((96, 64)).astype(np.float32)
    lab = rng.integers(0, 80, 96).astype(np.int32)
    idx, sim, hit = ops.retrieval_topk(torch.from_numpy(txt).to(DEV), torch.from_numpy(shp).to(DEV), torch.from_numpy(lab).to(DEV), 5)
    sims = txt.astype(np.float64) @ shp.astype(np.float64).T
    order = np.flip(np.argsort(sims, axis=1, kind="stable"), 1)
    np.testing.assert_array_equal(idx.cpu().numpy(), order[:, :5])
    np.testing.assert_array_equal(hit.cpu().numpy(), np.argmax(order == lab[:, None], axis=1))
    np.testing.assert_allclose(sim.cpu().numpy(), np.take_along_axis(sims, order[:, :5], 1), rtol=1e-13, atol=1e-13)


def test_u8_input_staging_is_bit_identical_to_the_host_prepared_inputs():
    """dense RGBA u8 grids and u8 renderings handed straight to the towers (mask / RGB/255 / CLIP Normalize
    derived on the device) give the very same embeddings as the reference's CPU-prepared COO batch and f32 images."""
    from tricolo_amd.data.synthetic import make_images_u8, normalise_images
    batch = syn.make_batch(4, voxel_size=32, num_views=None, seed=syn.BASE_SEED + 41, keep_grids=True)
    enc = SparseCNNEncoder(32, 32, 512, 512, precision="bf16x3")
    fill_module(enc)
    enc = enc.to(DEV).eval()
    with torch.no_grad():
        a = enc({k: v.to(DEV) for k, v in batch["voxels"].items()}, 4)
        b = enc({"rgba": batch["voxel_grids_u8"].to(DEV)}, 4)
    assert torch.equal(a, b)
    dense, mask = ops.voxel_from_rgba(batch["voxel_grids_u8"].to(DEV))
    dense2, mask2 = ops.voxel_scatter(batch["voxels"]["locs"].to(DEV), batch["voxels"]["feats"].to(DEV), 4, 32)
    assert torch.equal(dense, dense2) and torch.equal(mask, mask2)
    rng = np.random.default_rng(5)
    u8 = torch.from_numpy(np.stack([make_images_u8(rng, 2, 64) for _ in range(3)])).view(6, 3, 64, 64)
    f32 = normalise_images(u8.numpy())
    x_a = ops.nchw3_to_nhwc4(f32.to(DEV))
    x_b = ops.nchw3_u8_to_nhwc4(u8.to(DEV))
    assert torch.equal(x_a, x_b)
    img = MVCNNEncoder(512, 512, "resnet18", 2, precision="bf16x3")
    fill_module(img)
    img = img.to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(img(f32.to(DEV)), img(u8.to(DEV)))


def test_plain_bf16_mode_stated_tolerance(golden):
    """bf16 operands (1 MFMA product): the documented bound is 1e-2 on the loss and 5e-3 on unit-norm embeddings."""
    g = golden("step_cfg4_tri")
    ops.set_default_precision("bf16")
    net, cfg = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128)
    batch = syn.batch_to_device(syn.make_batch(8, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4), DEV)
    emb = net(batch)
    losses = net._calculate_losses(emb, "train_loss")
    diffs = {k: abs(v.item() - float(g[f"step0/{k}"])) for k, v in losses.items()}
    edif = {k: float(np.abs(v.detach().cpu().numpy() - g[f"emb/{k}"]).max()) for k, v in emb.items()}
    print("bf16 mode loss diffs", diffs, "embedding max diffs", edif)
    assert max(diffs.values()) < 1e-2 and max(edif.values()) < 5e-3


@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_f16_mode_meets_the_1e3_parity_bound(golden, case):
    """The f16 mode (f16 activation storage + f16 MFMA operands, fp32 accumulation, bf16x3 heads / GRU) against the golden
    vectors of the REAL fp32 reference: step-0 pair losses, total loss and unit-norm embeddings within 1e-3 - the north-star
    bound - on every BASELINE config that has a fixture.  This is the mode bench.py quotes as `value`."""
    tag, text, image, voxel, V, nv, S, B, seed_off, clip_text = case
    g = golden(f"step_{tag}")
    ops.set_default_precision("f16")
    net, cfg = _build_net(text, image, voxel, V, nv or 6, S)
    batch = syn.batch_to_device(syn.make_batch(B, voxel_size=V if voxel else None, num_views=nv if image else None, image_size=S,
                                               clip_text=clip_text, seed=syn.BASE_SEED + seed_off), DEV)
    opt = net.configure_optimizers()
    emb = net(batch)
    for k in emb:
        emb[k].retain_grad()
    losses = net._calculate_losses(emb, "train_loss")
    diffs = {k: abs(v.item() - float(g[f"step0/{k}"])) for k, v in losses.items()}
    edif = {k: float(np.abs(v.detach().cpu().numpy() - g[f"emb/{k}"]).max()) for k, v in emb.items()}
    losses["train_loss/total_loss"].backward()
    gdif = {k: float(np.abs(v.grad.cpu().numpy() - g[f"demb/{k}"]).max()) for k, v in emb.items()}
    for p in net.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    opt.step()
    after = net._calculate_losses(net(batch), "train_loss")["train_loss/total_loss"].item()
    _report(f"f16/{tag}", {"loss_abs_diff": diffs, "embedding_max_abs_diff": edif, "dloss_dembedding_max_abs_diff": gdif,
                           "loss_after_1_step": after, "reference_after_1_step": float(g["step1/total_loss"])})
    assert max(diffs.values()) < 1e-3, diffs
    assert max(edif.values()) < 1e-3, edif
    assert max(gdif.values()) < 1e-3, gdif
    assert abs(after - float(g["step1/total_loss"])) < 2e-2


@pytest.mark.parametrize("prec,gbound,smult", [("bf16", 1e-1, 8.0), ("f16", 3e-2, 10.0)])
def test_16bit_modes_backward_on_bi_v(golden, prec, gbound, smult):
    """BASELINE config 2 names bf16 on Bi(V): backward of the 16-bit storage paths at module level against the fp32 reference's
    parameter gradients.  Stated bounds (measured, profiles/r2/parity_report.json): gradient norms within 10 % (bf16: 6.3 %) /
    3 % (f16: 1.5 %), sampled entries within 0.8 rms (bf16: 0.50) / 0.3 rms (f16: 0.22) - operand rounding moves the max-pool /
    ReLU routing of the five pooled levels, so single entries move far more than the norms; the loss itself within 1e-2 / 1e-3."""
    g = golden("step_cfg1_biV")
    ops.set_default_precision(prec)
    net, cfg = _build_net("BiGRUEncoder", None, "SparseCNNEncoder", 32, 6, 128)
    batch = syn.batch_to_device(syn.make_batch(8, voxel_size=32, num_views=None, seed=syn.BASE_SEED + 1), DEV)
    emb = net(batch)
    losses = net._calculate_losses(emb, "train_loss")
    losses["train_loss/total_loss"].backward()
    _check_grads(net, g, "", rtol=gbound, sample_mult=smult, tag=f"biV_{prec}")
    ldiff = abs(losses["train_loss/total_loss"].item() - float(g["step0/train_loss/total_loss"]))
    _report(f"biV_{prec}/loss_abs_diff", ldiff)
    assert ldiff < (1e-3 if prec == "f16" else 1e-2)


def test_fused_adam_state_dict_round_trip_and_device_lr():
    """optimizer state must survive save -> load (Lightning `optimizer_states`), in torch.optim.Adam's own format,
    and a learning-rate change must reach a step that is replayed from a HIP graph."""
    from tricolo_amd.optim import FusedAdam
    torch.manual_seed(0)
    ws = [torch.randn(64, 32), torch.randn(128), torch.randn(16, 8, 4)]
    grads = [[torch.randn_like(w) for w in ws] for _ in range(6)]

    def make(cls):
        ps = [torch.nn.Parameter(w.clone().to(DEV)) for w in ws]
        return ps, cls(ps, lr=3.5e-4, weight_decay=1e-6)

    def run(ps, opt, steps):
        for gs in steps:
            for p, gr in zip(ps, gs):
                p.grad = gr.clone().to(DEV)
            opt.step()

    pt, ot = make(torch.optim.Adam)
    pf, of = make(FusedAdam)
    run(pt, ot, grads[:3]); run(pf, of, grads[:3])
    sd = of.state_dict()
    assert all(set(st) == {"step", "exp_avg", "exp_avg_sq"} for st in sd["state"].values())
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    # FusedAdam -> torch.optim.Adam and back, then three more steps with a decayed learning rate
    p2, o2 = make(torch.optim.Adam)
    p3, o3 = make(FusedAdam)
    with torch.no_grad():
        for a, b, c in zip(p2, p3, pf):
            a.copy_(c); b.copy_(c)
    o2.load_state_dict(sd)
    o3.load_state_dict(ot.state_dict())
    for o in (ot, o2, o3, of):
        o.param_groups[0]["lr"] = 1e-4
    run(pt, ot, grads[3:]); run(p2, o2, grads[3:]); run(p3, o3, grads[3:]); run(pf, of, grads[3:])
    for a, b, c, d in zip(pt, p2, p3, pf):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), atol=3e-7)
        np.testing.assert_allclose(c.detach().cpu().numpy(), a.detach().cpu().numpy(), atol=3e-7)
        np.testing.assert_allclose(d.detach().cpu().numpy(), a.detach().cpu().numpy(), atol=3e-7)
    # parameters without a gradient are skipped like torch does (no weight decay, no moment decay)
    pa, oa = make(torch.optim.Adam)
    pb, ob = make(FusedAdam)
    for ps, o in ((pa, oa), (pb, ob)):
        for p, gr in zip(ps, grads[0]):
            p.grad = gr.clone().to(DEV)
        ps[1].grad = None
        o.step()
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), atol=3e-7)
    # graph replay follows the device-side learning rate
    pg, og = make(FusedAdam)
    og.prepare()
    for p, gr in zip(pg, grads[0]):
        p.grad = gr.clone().to(DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        og.step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        og.step()
    pr, orf = make(torch.optim.Adam)
    for p, gr in zip(pr, grads[0]):
        p.grad = gr.clone().to(DEV)
    orf.step()                                                   # (the capture itself executes nothing)
    og.param_groups[0]["lr"] = orf.param_groups[0]["lr"] = 7e-5
    og.sync_lr()
    graph.replay()
    orf.step()
    for a, b in zip(pr, pg):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), atol=3e-7)


@pytest.mark.parametrize("prec", ["f16", "bf16x3"])
def test_config5_full_per_gpu_batch_properties(prec):
    """BASELINE config 5 at its real per-GPU size (64 samples: 64^3 voxels, 12 x 224^2 views, CLIP-text MLP): no fixture can
    hold this, so size-independent properties are checked - one full step runs (no TRI_ERR_UNSUPPORTED from the 32-bit buffer
    offset guards: the largest tensors here are 1.2 GB in f16 / 2.5 GB in fp32 storage), embeddings of the two towers that
    normalise are unit rows, every loss and gradient is finite, the loss is below the untrained bound 3 * log(64) and falls
    after the update."""
    ops.set_default_precision(prec)
    net, cfg = _build_net("CLIPTextEncoder", "MVCNNEncoder", "SparseCNNEncoder", 64, 12, 224)
    batch = syn.batch_to_device(syn.make_batch(64, voxel_size=64, num_views=12, image_size=224, clip_text=True, seed=syn.BASE_SEED + 55), DEV)
    opt = net.configure_optimizers()
    emb = net(batch)
    for k in ("image_features", "voxel_features"):
        assert emb[k].shape == (64, 512)
        np.testing.assert_allclose(emb[k].detach().norm(dim=1).cpu().numpy(), 1.0, atol=1e-4)
    losses = net._calculate_losses(emb, "train_loss")
    total = losses["train_loss/total_loss"]
    assert all(np.isfinite(v.item()) for v in losses.values()) and total.item() < 3 * np.log(64) + 1.0
    total.backward()
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    opt.step()
    after = net._calculate_losses(net(batch), "train_loss")["train_loss/total_loss"].item()
    _report(f"config5_b64/{prec}", {"loss": total.item(), "loss_after_1_step": after, "max_memory_GB": torch.cuda.max_memory_allocated() / 1e9})
    assert np.isfinite(after) and after < total.item()


@pytest.mark.timeout(900)
def test_config5_voxel_tower_at_full_size_against_the_oracle_forward():
    """the 64^3 x 64 kernel plans of the voxel tower - the level-0 brick kernel with 128-run bricks, level 1 on the
    row-list kernel, 128-wide split-K tiles on levels 2-4 - were only ever checked by properties at that size; the oracle comparison
    for 64^3 ran at batch 2, which plans differently.  Here the whole tower runs at config 5's real per-GPU batch and its embeddings are
    compared with the fp32 CPU oracle's forward on the same 64 samples (train-mode BatchNorm over the whole batch; ~25 s of CPU conv3d):
    bf16x3 within 2e-4, f16 within 1e-3 (the north star's bound) per element of the unit-norm rows, and the running statistics agree."""
    from oracle.modules import SparseCNNRef
    B, V = 64, 64
    batch = syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + 56)
    ref = SparseCNNRef(V, 32, 512, 512)
    fill_module(ref, prefix="voxel_encoder.")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        zr = ref(batch["voxels"], B)
    vox = {k: v.to(DEV) for k, v in batch["voxels"].items()}
    out = {}
    for prec, tol in (("bf16x3", EMB_TOL), ("f16", 1e-3)):
        ops.set_default_precision(prec)
        m = SparseCNNEncoder(V, 32, 512, 512)
        fill_module(m, prefix="voxel_encoder.")
        m = m.to(DEV)
        with torch.no_grad():
            m.train()
            z = m(vox, B)
        d = float((z.cpu() - zr).abs().max())
        out[prec] = d
        assert d <= tol, (prec, d)
        for name, v in m.state_dict().items():
            if "running_var" in name or "running_mean" in name:
                np.testing.assert_allclose(v.cpu().numpy(), ref.state_dict()[name].numpy(), rtol=2e-3, atol=2e-4, err_msg=f"{prec} {name}")
    _report("fullbatch/config5_voxel_tower_vs_oracle", {"max_abs_embedding_diff": out})


@pytest.mark.timeout(1800)
def test_config5_image_tower_at_full_size_against_the_oracle_forward():
    """the image tower at config 5's real per-GPU size - 64 samples x 12 views = 768 images of 224^2, where layer3 /
    layer4 run the 14- / 7-wide plans of conv_dma_kernel<128> and the krow / halo kernels their 56- / 28-wide forms - compared with the fp32
    CPU oracle's forward on the same images (train-mode BatchNorm over all 768 images): embeddings of the unit-norm rows within 2e-4
    (bf16x3) / 1e-3 (f16, the north star's bound), running statistics of the stem and of layer4's last BatchNorm agree."""
    from oracle.modules import MVCNNRef
    B, nv, S = 64, 12, 224
    batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 57)
    ref = MVCNNRef(512, 512, "resnet18", nv)
    fill_module(ref, prefix="image_encoder.")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    images = batch["images"].flatten(end_dim=1)
    with torch.no_grad():
        zr = ref(images, None)
    img = images.to(DEV)
    out = {}
    for prec, tol in (("bf16x3", EMB_TOL), ("f16", 1e-3)):
        ops.set_default_precision(prec)
        m = MVCNNEncoder(512, 512, "resnet18", nv)
        fill_module(m, prefix="image_encoder.")
        m = m.to(DEV)
        with torch.no_grad():
            m.train()
            z = m(img, None)
        d = float((z.cpu() - zr).abs().max())
        out[prec] = d
        assert d <= tol, (prec, d)
        sd, rsd = m.state_dict(), ref.state_dict()
        for name in ("net_1.1.running_mean", "net_1.1.running_var", "net_1.7.1.bn2.running_mean", "net_1.7.1.bn2.running_var"):
            np.testing.assert_allclose(sd[name].cpu().numpy(), rsd[name].numpy(), rtol=3e-3, atol=3e-4, err_msg=f"{prec} {name}")
        del m
        torch.cuda.empty_cache()
    _report("fullbatch/config5_image_tower_vs_oracle", {"max_abs_embedding_diff": out})


@pytest.mark.timeout(900)
def test_f16_overflow_guard_never_fires_on_the_bench_workload():
    """2,000 f16 training steps of the bench workload (BASELINE config 4 per-GPU shard, 8 resident synthetic batches
    as bench.py replays them, static 2^12 gradient scale) - the per-step non-finite guard must not skip a single step and the loss must
    stay finite and fall."""
    ops.set_default_precision("f16")
    net, cfg = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128)
    opt = net.configure_optimizers()
    batches = [syn.batch_to_device(syn.make_batch(32, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4 + i), DEV) for i in range(8)]
    first = last = None
    for step in range(2000):
        opt.zero_grad(set_to_none=True)
        loss = net.training_step(batches[step % 8], step)
        loss.backward()
        opt.step()
        if step == 0:
            first = float(loss.item())
        if step % 500 == 499:
            last = float(loss.item())
            assert np.isfinite(last), (step, last)
    assert opt.skipped_steps() == 0 and opt.nonfinite_skipped() == 0, (opt.skipped_steps(), opt.nonfinite_skipped())
    assert last < first, (first, last)
    _report("f16_guard_2000_steps", {"first_loss": first, "last_loss": last, "skipped_steps": 0})


def test_replayed_bench_steps_are_bit_reproducible():
    """Round 6: the bench workload's step captured into a HIP graph and replayed, THREE independent runs from the same initial weights:
    every loss of the trajectory must agree bit for bit.  A kernel that clobbers a register behind an un-waited asynchronous load, or a
    side branch racing the chain it was forked from, shows up as a run whose trajectory leaves the others' after a few dozen steps (this
    caught conv_halo_rows_kernel's first L2 warm-up, whose loads had register destinations: 5-10 % of bench runs ended on another loss)."""
    ops.set_default_precision("f16")
    from tricolo_amd import parallel
    batches = [syn.batch_to_device(syn.make_batch(32, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4 + i), DEV) for i in range(4)]
    runs = []
    for rep in range(3):
        net, cfg = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128)
        opt = net.configure_optimizers()
        opt.prepare(captures=len(batches))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                                # eager warm-up on a side stream (lazy buffers), then back to the initial state
            snap = [p.detach().clone() for p in net.parameters()]
            bufs = [b.detach().clone() for b in net.buffers()]
            parallel.dp_training_step(net, batches[0], opt)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p_, q in zip(net.parameters(), snap):
                p_.copy_(q)
            for b_, q in zip(net.buffers(), bufs):
                b_.copy_(q)
            opt._flat_m.zero_(); opt._flat_v.zero_(); opt._step_dev.zero_()
        graphs, outs = [], []
        pool = None
        for b in batches:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                outs.append(parallel.dp_training_step(net, b, opt)["train_loss/total_loss"])
            pool = pool or g.pool()
            graphs.append(g)
        losses = []
        for step in range(60):
            graphs[step % len(graphs)].replay()
            losses.append(outs[step % len(graphs)].clone())
        torch.cuda.synchronize()
        runs.append(torch.stack(losses).cpu())
        del graphs, outs, net, opt
        torch.cuda.empty_cache()
    assert torch.isfinite(runs[0]).all()
    for r in runs[1:]:
        assert torch.equal(r, runs[0]), f"trajectories differ from step {int((r != runs[0]).nonzero()[0])}: {r[-1].item()} vs {runs[0][-1].item()}"


@pytest.mark.parametrize("tag,text,image,voxel,V,nv,S,B", [
    ("config2", "BiGRUEncoder", None, "SparseCNNEncoder", 32, 6, 128, 64),
    ("config3", "BiGRUEncoder", "MVCNNEncoder", None, 32, 6, 128, 64),
    ("config4", "BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128, 32),
], ids=["config2", "config3", "config4"])
def test_full_per_gpu_batch_properties_and_mode_agreement(tag, text, image, voxel, V, nv, S, B):
    """BASELINE configs 2-4 at their real per-GPU batch (the fixtures hold 8 samples).  Every precision mode - bf16x3, f16 and the
    bf16 that BASELINE config 2 names - runs one step on the same weights and batch; its step-0 loss and embeddings are compared with
    the fp32 CPU ORACLE's forward on that full-size batch (not only with another HIP mode): bf16x3 and f16 inside the north star's
    1e-3 on the loss / 5e-4 on the unit-norm embeddings, bf16 inside its stated 1e-2 / 5e-3.  Plus the size-independent properties of
    test_config5_full_per_gpu_batch_properties.  These sizes also take kernel variants the fixtures never reach (per-workgroup
    BatchNorm records, the slab-based layer1 weight gradient of config 3, row-list launches with > 10^5 rows, the brick kernel)."""
    from itertools import combinations
    from oracle import modules as om
    host_batch = syn.make_batch(B, voxel_size=V if voxel else None, num_views=nv if image else None, image_size=S, seed=syn.BASE_SEED + 61)
    ref = om.TriCoLoRef(om.BiGRURef(syn.DEFAULT_VOCAB, 512), om.MVCNNRef(512, 512, "resnet18", nv) if image else None,
                        om.SparseCNNRef(V, 32, 512, 512) if voxel else None)
    fill_module(ref)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():                                   # train-mode forward (batch statistics), fp32, no step
        remb = ref(host_batch)
        rtotal = float(sum(om.nt_xent_ref(remb[a], remb[b], 0.1, 0.25) for a, b in combinations(remb.keys(), 2)))
    bounds = {"bf16x3": (1e-3, 5e-4), "f16": (1e-3, 5e-4), "bf16": (1e-2, 5e-3)}
    res, rep = {}, {"loss_oracle_fp32": rtotal}
    for prec in ("bf16x3", "f16", "bf16"):
        ops.set_default_precision(prec)
        torch.manual_seed(1234)
        net, cfg = _build_net(text, image, voxel, V, nv, S)
        batch = syn.batch_to_device(host_batch, DEV)
        opt = net.configure_optimizers()
        emb = net(batch)
        for k, v in emb.items():
            assert v.shape == (B, 512)
            np.testing.assert_allclose(v.detach().norm(dim=1).cpu().numpy(), 1.0, atol=1e-4)
        total = net._calculate_losses(emb, "train_loss")["train_loss/total_loss"]
        total.backward()
        for name, p in net.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
        opt.step()
        after = net._calculate_losses(net(batch), "train_loss")["train_loss/total_loss"].item()
        assert np.isfinite(after) and after < total.item()
        res[prec] = (total.item(), after, {k: v.detach().float().cpu() for k, v in emb.items()})
        dl = abs(total.item() - rtotal)
        de = max(float((res[prec][2][k] - remb[k]).abs().max()) for k in remb)
        rep.update({f"loss_{prec}": total.item(), f"loss_abs_diff_vs_oracle_{prec}": dl, f"embedding_max_abs_diff_vs_oracle_{prec}": de,
                    f"loss_after_1_step_{prec}": after})
        _report(f"fullbatch/{tag}", rep)
        assert dl < bounds[prec][0] and de < bounds[prec][1], (prec, dl, de)
        del net, opt
    dl = abs(res["f16"][0] - res["bf16x3"][0])
    de = max(float((res["f16"][2][k] - res["bf16x3"][2][k]).abs().max()) for k in res["f16"][2])
    rep.update({"loss_abs_diff_f16_vs_bf16x3": dl, "embedding_max_abs_diff_f16_vs_bf16x3": de})
    _report(f"fullbatch/{tag}", rep)
    assert dl < 1e-3 and de < 5e-4


_HELDOUT_CACHE = {}


def _heldout_batches(steps):
    """Training batches (fixed order, resident on the GPU) and the held-out items of oracle/make_heldout_rr.py, built once."""
    if "data" not in _HELDOUT_CACHE:
        from oracle import make_heldout_rr as mk
        train, held = mk.datasets()
        rng = np.random.default_rng(mk.SEED_ORDER)
        batches = [syn.batch_to_device(syn.collate_items([train[i] for i in mk.batch_indices(rng, mk.TRAIN_SHAPES)], voxel=True, views=True), DEV)
                   for _ in range(steps)]
        _HELDOUT_CACHE["data"] = (mk, train, held, batches)
    return _HELDOUT_CACHE["data"]


@pytest.mark.parametrize("prec,bound", [("bf16x3", 0.2), ("f16", 0.2), ("bf16", 1.0)])
def test_heldout_retrieval_rr1(golden, prec, bound):
    """North star: retrieval RR@1 within +-0.2 of the reference on the same held-out synthetic set (512 unseen shapes x
    5 captions = 2,560 queries of a learnable factor space).  The reference side is the REAL reference TriCoLoNet (under the import
    shims of oracle/make_golden.py) trained in the build container and scored by the REAL compute_metrics
    (oracle/make_heldout_rr.py -> tests/golden/heldout_rr.npz, `source` = "reference"); the HIP path is trained here from the same recipe weights on the
    same batches in the same order, embedded in eval mode, and ranked by the device retrieval kernel through compute_metrics.
    0.2 points = 5 queries of 2,560.  The bf16 throughput mode is reported with its own stated bound."""
    from tricolo_amd.evaluation.eval_retrieval import compute_metrics
    g = golden("heldout_rr")
    assert str(g["source"]) == "reference"
    cps = [int(c) for c in g["checkpoints"]]
    steps = int(os.environ.get("TRICOLO_HELDOUT_STEPS", cps[-1]))
    cps = [c for c in cps if c <= steps]
    assert cps, "TRICOLO_HELDOUT_STEPS is below the first checkpoint of the fixture"
    mk, train, held, batches = _heldout_batches(cps[-1])
    assert mk.data_sha(train) == str(g["train_sha"]) and mk.data_sha(held) == str(g["held_sha"])
    ops.set_default_precision(prec)
    net, cfg = _build_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", mk.V, mk.NV, mk.S)
    opt = net.configure_optimizers()
    report = {}
    for step in range(1, cps[-1] + 1):
        opt.zero_grad(set_to_none=True)
        loss = net.training_step(batches[step - 1], step)
        loss.backward()
        opt.step()
        if step in cps:
            net.eval()
            e = mk.embed(net, held, device=DEV)
            net.train()
            shape = e["image"] + e["voxel"]                                   # tricolo_net.py:134-139
            tuples = [(None, "synthetic", f"shape{it['shape']:05d}", e["text"][i], shape[i]) for i, it in enumerate(held)]
            m = compute_metrics("Synthetic", {"caption_embedding_tuples": tuples})
            ref_rr = g[f"cp{step}/recall_rate"]
            report[str(step)] = {"RR@1": 100 * float(m["recall_rate"][0]), "reference_RR@1": 100 * float(ref_rr[0]),
                                 "RR@5": 100 * float(m["recall_rate"][4]), "reference_RR@5": 100 * float(ref_rr[4]),
                                 "delta_RR@1": 100 * float(m["recall_rate"][0] - ref_rr[0]),
                                 "top1_index_agreement": float(np.mean(m["indices"][:, 0] == g[f"cp{step}/indices"][:, 0])),
                                 "train_loss": float(loss.item()), "reference_train_loss": float(g["losses"][step - 1])}
    _report(f"heldout_rr/{prec}", report)
    print(prec, report)
    last = report[str(cps[-1])]
    assert last["RR@1"] > 50.0                                                # learnable on unseen shapes (chance: 0.2)
    assert abs(last["delta_RR@1"]) <= bound + 1e-9, report
    # every checkpoint is bounded, not only the last.  Two trainings from identical weights diverge chaotically (Adam's
    # sign-like steps amplify last-bit differences): on the steep part of the curve (step 200: RR@1 ~ 89.5 %) the fp32-grade bf16x3 mode
    # itself sits 0.1-0.3 points and 3.8 % of the top-1 indices away from the reference, f16 has been measured between -0.59 and +0.31.
    # The bounds follow the slope: |delta RR@1| <= 1.0 / 0.5 points at steps 200 / 400 (one query = 0.039), the north star's 0.2 at the end;
    # top-1 index agreement >= 94 % / 97.5 % / 99 % (bf16, outside the 1e-3 mode: 2 x the point bounds, same agreement floors).
    scale = 1.0 if prec != "bf16" else 2.0
    for cp, dmax, agree in ((200, 1.0, 0.94), (400, 0.5, 0.975), (600, None, 0.99)):
        r = report.get(str(cp))
        if r is None:
            continue
        if dmax is not None:
            assert abs(r["delta_RR@1"]) <= scale * dmax, (cp, report)
        assert r["top1_index_agreement"] >= agree, (cp, report)


def test_cpu_input_fails_loudly():
    m = SparseCNNEncoder(32, 32, 512, 512)
    batch = syn.make_batch(2, voxel_size=32, num_views=None, seed=1)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(batch["voxels"], 2)
