"""Per-tensor gradient deviation of the HIP image tower from the float64 reference gradients (tests/golden/mvcnn_sens.npz)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.recipe import fill_module, probe
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
from tricolo_amd.model.module.img_encoder.mv_cnn import MVCNNEncoder
g = dict(np.load("tests/golden/mvcnn_sens.npz"))
for prec in sys.argv[1:] or ["bf16x3"]:
    ops.set_default_precision(prec)
    m = MVCNNEncoder(512, 512, "resnet18", 6); fill_module(m, prefix="image_encoder."); m = m.cuda()
    batch = syn.make_batch(8, voxel_size=None, num_views=6, image_size=128, seed=syn.BASE_SEED + 3)
    z = m(batch["images"].flatten(end_dim=1).cuda(), batch)
    up = torch.randn((8, 512), generator=torch.Generator().manual_seed(13))
    (z * up.cuda()).sum().backward()
    print("==", prec, "z max diff", float(np.abs(z.detach().cpu().numpy() - g["v6s128/z64"]).max()))
    rows = []
    for name, p in m.named_parameters():
        rn = float(g[f"v6s128/gradnorm64/{name}"]); sn = float(g[f"v6s128/sens_norm/{name}"]); ss = float(g[f"v6s128/sens_sample/{name}"])
        n, smp = probe(p.grad.cpu()); rms = rn / np.sqrt(p.numel())
        ds = float(np.abs(smp.astype(np.float64) - g[f"v6s128/gradsample64/{name}"]).max()) / rms
        rows.append((abs(n - rn) / rn, sn, ds, ss, name))
    for r in rows: print("dn %.2e sens %.2e | ds %.2e sens %.2e  %s" % r)
