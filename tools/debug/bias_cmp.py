import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.recipe import fill_module
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
from tricolo_amd.model.module.img_encoder.mv_cnn import MVCNNEncoder
ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bias64.npz")))
ops.set_default_precision("bf16x3")
m = MVCNNEncoder(512, 512, "resnet18", 6); fill_module(m, prefix="image_encoder."); m = m.cuda()
batch = syn.make_batch(8, voxel_size=None, num_views=6, image_size=128, seed=syn.BASE_SEED + 3)
z = m(batch["images"].flatten(end_dim=1).cuda(), batch)
up = torch.randn((8, 512), generator=torch.Generator().manual_seed(13))
(z * up.cuda()).sum().backward()
g = {n: p.grad.double().cpu().numpy() for n, p in m.named_parameters()}
for n in sorted(k[4:] for k in ref if k.startswith("f64/")):
    a, b, h = ref["f64/" + n], ref["f32/" + n], g[n]
    e = h - a
    cos = float(np.dot(e, a) / (np.linalg.norm(e) * np.linalg.norm(a) + 1e-300))
    print(f"{n:34s} |hip-f64|/|f64| {np.linalg.norm(e)/np.linalg.norm(a):.2e}  |f32-f64|/|f64| {np.linalg.norm(b-a)/np.linalg.norm(a):.2e}  cos(err,grad) {cos:+.2f}  "
          f"max elem err/rms {np.abs(e).max()/ (np.linalg.norm(a)/np.sqrt(a.size)):.2e}")
n = "net_1.5.0.bn2.bias"
a, h = ref["f64/" + n], g[n]
idx = np.argsort(-np.abs(h - a))[:8]
print("worst channels", idx, (h - a)[idx], a[idx])
