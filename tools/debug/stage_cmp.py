"""Stage-by-stage comparison of the HIP image tower with the float64 oracle: block outputs, pooled features, view arg-max, and the
gradient entering the trunk."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import modules as om
from oracle.recipe import fill_module
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
from tricolo_amd.model.module.img_encoder.mv_cnn import MVCNNEncoder
B, nv, S = 8, 6, 128
batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
up = torch.randn((B, 512), generator=torch.Generator().manual_seed(13))
torch.set_num_threads(16)
ref = om.MVCNNRef(512, 512, "resnet18", nv); fill_module(ref, prefix="image_encoder."); ref = ref.double()
acts = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); acts[name] = out
    return f
for li in (4, 5, 6, 7):
    for bi in (0, 1):
        ref.net_1[li][bi].register_forward_hook(hook(f"{li}.{bi}"))
zr = ref(batch["images"].flatten(end_dim=1).double(), {}); (zr * up.double()).sum().backward()
feat = acts["7.1"]                                      # [N,512,4,4]
pooled_r = feat.mean(dim=(2, 3)).view(B, nv, 512)
arg_r = pooled_r.argmax(dim=1)
srt = pooled_r.sort(dim=1, descending=True).values
gap = (srt[:, 0] - srt[:, 1]) / srt[:, 0].abs().clamp_min(1e-30)
print("oracle: relative gap between best and second view: min %.2e  p1 %.2e  p10 %.2e  median %.2e; exact ties %d" % (
    gap.min(), gap.flatten().kthvalue(max(1, int(0.01 * gap.numel()))).values, gap.flatten().kthvalue(int(0.1 * gap.numel())).values, gap.median(),
    int((gap == 0).sum())))
ops.set_default_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
m = MVCNNEncoder(512, 512, "resnet18", nv); fill_module(m, prefix="image_encoder."); m = m.cuda()
with torch.no_grad():
    z, saved = m._forward_impl(batch["images"].flatten(end_dim=1).cuda(), save=True)
blocks = saved["lower"]["blocks"] + saved["upper"]["blocks"]
names = ["4.0", "4.1", "5.0", "5.1", "6.0", "6.1", "7.0", "7.1"]
for nme, sv in zip(names, blocks):
    out = sv[-1].double().cpu()                          # [N,1,H,W,C]
    r = acts[nme].detach().permute(0, 2, 3, 1).unsqueeze(1)
    print(f"block {nme} output rel L2 err {float((out - r).norm() / r.norm()):.2e}  max abs {float((out - r).abs().max()):.2e}")
pooled_h = saved["upper"]["pooled"].double().cpu()
arg_h = saved["upper"]["arg"].cpu()
pr = pooled_r.max(dim=1).values
print("pooled rel L2 err %.2e; argmax view differs in %d of %d (b, c) pairs" % (float((pooled_h - pr).norm() / pr.norm()), int((arg_h.long() != arg_r).sum()), arg_r.numel()))
bad = (arg_h.long() != arg_r)
if bad.any():
    print("   gaps at the disagreeing pairs:", gap[bad][:10].tolist())
