"""Dense-masked restatement of the five spconv classes the reference voxel tower uses.  TEST INFRASTRUCTURE.

Call sites restated: /root/reference/tricolo/model/module/voxel_encoder/sparse_cnn.py:11-37 (SparseSequential of
SubMConv3d / BatchNorm1d / ReLU / SparseMaxPool3d / ToDense) and :47 (SparseConvTensor).

spconv itself (``spconv-cu113``, un-pinned, /root/reference/setup.py:11) is not under /root/reference and cannot
be installed here: PARITY UNPINNED for this file.  Semantics follow SURVEY.md section 8(a) "Semantic spec for a-4":
  * SubMConv3d(k=3, bias=False): y[b,p] = m[b,p] * sum_{d in {-1,0,1}^3} W[d] . x[b,p+d]   (cross-correlation,
    zero outside the grid / at inactive sites, outputs only at active sites); weight stored [Cout,kd,kh,kw,Cin].
  * BatchNorm1d on the [N_active, C] feature matrix: train-mode statistics over active sites of the whole batch.
  * SparseMaxPool3d(2,2): output site active iff any child active; value = max over active children.
  * ToDense: [B, C, D, H, W] channels-first, zeros at inactive sites.
The same classes double as the ``spconv.pytorch`` shim under which oracle/make_golden.py imports the real
``SparseCNNEncoder`` wrapper class.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class SparseConvTensor:
    """Dense-masked stand-in: ``dense`` [B,C,D,H,W] (zero where inactive), ``mask`` [B,1,D,H,W] in {0,1}."""

    def __init__(self, features, indices, spatial_shape, batch_size):
        D, H, W = (int(s) for s in spatial_shape)
        C = features.shape[1]
        idx = indices.long()
        dense = features.new_zeros((batch_size, D, H, W, C))
        mask = features.new_zeros((batch_size, D, H, W, 1))
        if idx.shape[0] > 0:
            dense = dense.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), features)
            mask = mask.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), features.new_ones((idx.shape[0], 1)))
        self.dense = dense.permute(0, 4, 1, 2, 3).contiguous()
        self.mask = mask.permute(0, 4, 1, 2, 3).contiguous()
        self.batch_size = batch_size

    @classmethod
    def from_dense(cls, dense, mask):
        obj = cls.__new__(cls)
        obj.dense, obj.mask, obj.batch_size = dense, mask, dense.shape[0]
        return obj

    @property
    def n_active(self):
        return int(self.mask.sum().item())


class SubMConv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=False, **kw):
        super().__init__()
        assert not bias and kernel_size == 3
        k = kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        bound = 1.0 / math.sqrt(in_channels * k ** 3)          # kaiming_uniform(a=sqrt(5)) on fan_in [ext]
        nn.init.uniform_(self.weight, -bound, bound)

    def forward(self, x: SparseConvTensor):
        w = self.weight.permute(0, 4, 1, 2, 3)                  # -> [Cout, Cin, kd, kh, kw]
        y = F.conv3d(x.dense, w, padding=1) * x.mask
        return SparseConvTensor.from_dense(y, x.mask)


class SparseMaxPool3d(nn.Module):
    def __init__(self, kernel_size, stride):
        super().__init__()
        assert kernel_size == 2 and stride == 2

    def forward(self, x: SparseConvTensor):
        # inputs are post-ReLU (>= 0) and exactly 0 at inactive sites, so a dense max is exact
        return SparseConvTensor.from_dense(F.max_pool3d(x.dense, 2, 2), F.max_pool3d(x.mask, 2, 2))


class ToDense(nn.Module):
    def forward(self, x: SparseConvTensor):
        return x.dense


def masked_batchnorm(bn: nn.BatchNorm1d, x: SparseConvTensor):
    """nn.BatchNorm1d applied to the [N_active, C] feature matrix, written back into the dense grid."""
    m = x.mask
    n = m.sum()
    if n.item() == 0:
        return x
    y = x.dense
    if bn.training:
        mean = (y * m).sum(dim=(0, 2, 3, 4)) / n
        cen = (y - mean.view(1, -1, 1, 1, 1)) * m
        var = (cen * cen).sum(dim=(0, 2, 3, 4)) / n
        with torch.no_grad():
            mom = bn.momentum
            bn.running_mean.mul_(1 - mom).add_(mom * mean.detach())
            bn.running_var.mul_(1 - mom).add_(mom * var.detach() * (n / (n - 1)))
            bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean, bn.running_var
    inv = torch.rsqrt(var + bn.eps)
    out = (y - mean.view(1, -1, 1, 1, 1)) * (inv * bn.weight).view(1, -1, 1, 1, 1) + bn.bias.view(1, -1, 1, 1, 1)
    return SparseConvTensor.from_dense(out * m, m)


class SparseSequential(nn.Sequential):
    def forward(self, x):
        for mod in self:
            if isinstance(mod, nn.BatchNorm1d):
                x = masked_batchnorm(mod, x)
            elif isinstance(mod, nn.ReLU):
                if isinstance(x, SparseConvTensor):
                    x = SparseConvTensor.from_dense(F.relu(x.dense), x.mask)
                else:
                    x = F.relu(x)
            else:
                x = mod(x)
        return x
