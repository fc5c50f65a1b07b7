import torch, sys
sys.path.insert(0, '.')
from tricolo_amd import ops
DEV = 'cuda'
M, C, store = 3072, 512, torch.float16
gen = torch.Generator().manual_seed(97)
ya, yb = torch.randn(M, C, generator=gen).to(DEV).to(store), (torch.randn(M, C, generator=gen) * 0.5 + 0.2).to(DEV).to(store)
out = torch.relu(torch.randn(M, C, generator=gen)).to(DEV).to(store)
dout = (torch.randn(M, C, generator=gen) * 64).to(DEV).to(store)
ga, gb = (torch.rand(C, generator=gen) + 0.5).to(DEV), (torch.rand(C, generator=gen) + 0.5).to(DEV)
coa, cob = ops.BNCoeffs(C, DEV), ops.BNCoeffs(C, DEV)
for co, y in ((coa, ya), (cob, yb)):
    yf = y.float(); co.mean.copy_(yf.mean(0)); co.invstd.copy_(1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5))
scale = 1.0 / 4096
d1 = dout.clone()
r1 = ops.bn_bwd(ya, d1, coa, ga, count_host=M, inplace=False, relu_out=out, g_masked=d1, out_scale=scale)
r2 = ops.bn_bwd(yb, d1, cob, gb, count_host=M, inplace=False, out_scale=scale)
d2 = dout.clone()
p = ops.bn_bwd_pair(ya, coa, ga, yb, cob, gb, d2, out, M, g_masked=d2, out_scale=scale)
torch.cuda.synchronize()
names = ["dya", "dga", "dba", "dyb", "dgb", "dbb"]
for n, a, b in zip(names, p, list(r1) + list(r2)):
    print(n, float((a.float() - b.float()).abs().max()), float(b.float().abs().max()))
print("gm", float((d1.float() - d2.float()).abs().max()))
