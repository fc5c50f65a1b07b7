"""Full-tensor relative L2 error of every HIP image-tower parameter gradient against the fp32 CPU oracle (same weights, same batch)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import modules as om
from oracle.recipe import fill_module
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
from tricolo_amd.model.module.img_encoder.mv_cnn import MVCNNEncoder
B, nv, S = int(os.environ.get("B", 8)), 6, int(os.environ.get("S", 128))
batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
up = torch.randn((B, 512), generator=torch.Generator().manual_seed(13))
torch.set_num_threads(16)
dt = torch.float64 if os.environ.get("REF64", "1") == "1" else torch.float32
ref = om.MVCNNRef(512, 512, "resnet18", nv); fill_module(ref, prefix="image_encoder."); ref = ref.to(dt)
zr = ref(batch["images"].flatten(end_dim=1).to(dt), {}); (zr * up.to(dt)).sum().backward()
rg = {n: p.grad.double() for n, p in ref.named_parameters()}
for prec in sys.argv[1:] or ["bf16x3"]:
    ops.set_default_precision(prec)
    m = MVCNNEncoder(512, 512, "resnet18", nv); fill_module(m, prefix="image_encoder."); m = m.cuda()
    z = m(batch["images"].flatten(end_dim=1).cuda(), batch)
    (z * up.cuda()).sum().backward()
    print("==", prec, "B", B, "z max diff", float((z.detach().cpu().double() - zr.detach().double()).abs().max()))
    for n, p in m.named_parameters():
        h = p.grad.double().cpu(); a = rg[n]
        print(f"  {n:34s} rel L2 err {float((h - a).norm() / a.norm()):.2e}   norm ratio {float(h.norm() / a.norm()):.5f}")
