"""Conditioning-aware gradient pin for the ResNet-18 tower.  TEST INFRASTRUCTURE.

The image tower's parameter gradients are ill-conditioned through train-mode BatchNorm, ReLU, max-pool and view-max routing: a
relative weight perturbation of 1.5e-5 (the size of the split-bf16 operand error of the HIP path) moves some gradient norms of the
fp32 CPU oracle by ~0.7 %.  A flat bound (2e-2 on norms, 0.3 rms on samples: tests/test_gpu_modules.py round 2) therefore cannot see
a 1 % bug in a well-conditioned tensor.  This script measures the conditioning PER TENSOR and stores it next to float64 gradients:

  * the REAL reference wrapper class (tricolo/model/module/img_encoder/mv_cnn.py:13-33 under the shims of oracle/make_golden.py, on
    the restated torchvision ResNet-18) in float64, recipe weights, the batch of tests/golden/mvcnn.npz [v6s128];
  * gradients of every parameter in float64: norm + the 16 probe entries;
  * TRIALS times: every parameter multiplied element-wise by 1 + 1.5e-5 * N(0, 1), gradients recomputed in float64; per tensor the
    largest relative norm change and the largest probe-entry change over the tensor's rms are the tensor's SENSITIVITY.

Round 3 finding behind the third quantity (`sens_l2`, the full-tensor relative L2 change): forward activations that differ by ~1e-5
relative flip the ReLU mask of the elements nearest zero; each flip changes one gradient element by O(rms), so a few dozen flips in a
4e5-element tensor move the gradient by percent in L2 while its NORM changes only at second order - norms and 16 probes cannot tell
that apart from a kernel bug, a full-vector comparison against a measured L2 sensitivity can.  The 40 BatchNorm weight / bias
gradients (<= 512 elements each) are therefore stored whole.

tests/test_gpu_modules.py::test_mvcnn_gradients_within_measured_conditioning bounds the HIP path's deviation from the float64
gradient by 3 x that sensitivity per tensor (plus a small floor).  Run in the build container (minutes of CPU):

    python -m oracle.make_mvcnn_sensitivity          # writes tests/golden/mvcnn_sens.npz
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import make_golden as mg  # noqa: E402
from oracle.recipe import fill_module, probe  # noqa: E402
from tricolo_amd.data import synthetic as syn  # noqa: E402

EPS, TRIALS = 1.5e-5, 4


def grads(m, images, up):
    m.zero_grad(set_to_none=True)
    z = m(images, {})
    (z * up).sum().backward()
    return z.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}


def main():
    mg.install_shims()
    torch.set_num_threads(int(os.environ.get("THREADS", "4")))
    from tricolo.model.module.img_encoder.mv_cnn import MVCNNEncoder
    tag, B, nv, S = "v6s128", 8, 6, 128
    m = MVCNNEncoder(z_dim=512, out_dim=512, cnn_name="resnet18", num_views=nv)
    fill_module(m, prefix="image_encoder.")
    m = m.double()
    batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
    images = batch["images"].flatten(end_dim=1).double()
    up = torch.randn((B, 512), generator=torch.Generator().manual_seed(13)).double()      # the upstream gradient of mvcnn.npz
    z0, g0 = grads(m, images, up)
    base = {n: p.detach().clone() for n, p in m.named_parameters()}
    out = {f"{tag}/input_sha": mg.sha(batch["images"]), f"{tag}/z64": z0.numpy(), "eps": np.float64(EPS), "trials": np.int64(TRIALS)}
    sens_n = {n: 0.0 for n in g0}
    sens_s = {n: 0.0 for n in g0}
    sens_l2 = {n: 0.0 for n in g0}                     # full-tensor relative L2 change: what a norm (second order) cannot see
    gen = torch.Generator().manual_seed(2025)
    for trial in range(TRIALS):
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(base[n] * (1.0 + EPS * torch.randn(p.shape, generator=gen, dtype=torch.float64)))
        _, g1 = grads(m, images, up)
        for n in g0:
            n0, s0 = probe(g0[n])
            n1, s1 = probe(g1[n])
            rms = max(n0 / np.sqrt(g0[n].numel()), 1e-30)
            idx = torch.from_numpy(__import__("oracle.recipe", fromlist=["sample_indices"]).sample_indices(g0[n].numel()))
            d = (g1[n].reshape(-1)[idx] - g0[n].reshape(-1)[idx]).abs().max().item()
            sens_l2[n] = max(sens_l2[n], float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-300)))
            sens_n[n] = max(sens_n[n], abs(n1 - n0) / max(n0, 1e-30))
            sens_s[n] = max(sens_s[n], d / rms)
        print(f"trial {trial}: worst norm sensitivity {max(sens_n.values()):.3e}, worst sample sensitivity {max(sens_s.values()):.3e}, "
              f"L2 sensitivity median {float(np.median(list(sens_l2.values()))):.3e} max {max(sens_l2.values()):.3e}", flush=True)
    for n in g0:
        nn_, _ = probe(g0[n])
        idx = torch.from_numpy(__import__("oracle.recipe", fromlist=["sample_indices"]).sample_indices(g0[n].numel()))
        out[f"{tag}/gradnorm64/{n}"] = np.float64(nn_)
        out[f"{tag}/gradsample64/{n}"] = g0[n].reshape(-1)[idx].numpy()                  # float64
        out[f"{tag}/sens_norm/{n}"] = np.float64(sens_n[n])
        out[f"{tag}/sens_sample/{n}"] = np.float64(sens_s[n])
        out[f"{tag}/sens_l2/{n}"] = np.float64(sens_l2[n])
        if g0[n].numel() <= 512:                       # BatchNorm / bias vectors: the whole float64 gradient (element-wise L2 check)
            out[f"{tag}/grad64/{n}"] = g0[n].reshape(-1).numpy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "mvcnn_sens.npz"), **out)
    order = sorted(sens_n, key=lambda k: -sens_n[k])
    for n in order[:8]:
        print(f"  {n:45s} norm sens {sens_n[n]:.2e}  sample sens {sens_s[n]:.2e}")
    print("median norm sens", float(np.median(list(sens_n.values()))))


if __name__ == "__main__":
    main()
