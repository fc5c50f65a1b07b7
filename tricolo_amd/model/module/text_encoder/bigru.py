"""BiGRUEncoder - drop-in for /root/reference/tricolo/model/module/text_encoder/bigru.py:8-18.

Same constructor (vocab_size, out_dim, **kwargs - `clip_model` ignored), same forward(tokens [B,L] int, data_dict)
-> [B, out_dim] unit rows, same state-dict keys (embedding_layer.weight, gru.{weight_ih,weight_hh,bias_ih,bias_hh}_l0
[_reverse], fc.{weight,bias}).  As in the reference there is no packing: pad tokens (id 0, zero embedding row) are
stepped through the GRU and the forward final state is taken after the trailing pads.

Round 1: embedding gather + the 96-step recurrence run through PyTorch-ROCm (nn.GRU -> MIOpen), which the task
allows for this tower; the output projection + tanh is the MFMA dense kernel and the
normalise is the row kernel.  The persistent fused recurrence kernel is the next row.
"""
import torch
import torch.nn as nn

from .... import ops
from ....layers import L2NormFn, LinearFn, TriModule, require_gpu


class BiGRUEncoder(TriModule):
    def __init__(self, vocab_size, out_dim, precision=None, **kwargs):
        super().__init__()
        self.embedding_layer = nn.Embedding(vocab_size, 256, padding_idx=0)
        self.gru = nn.GRU(input_size=256, hidden_size=128, num_layers=1, bidirectional=True)
        self.fc = nn.Linear(256, out_dim)
        self.precision = precision

    def forward(self, x, data_dict=None):
        require_gpu(x, "BiGRUEncoder")
        emb = torch.transpose(self.embedding_layer(x), 0, 1)                   # bigru.py:15
        h0 = torch.zeros((2, emb.shape[1], 128), dtype=torch.float32, device=emb.device)
        _, hidden = self.gru(emb, h0)                                          # bigru.py:17
        feat = torch.cat((hidden[-2], hidden[-1]), dim=1)
        prec = self.precision or ops.default_precision()
        return L2NormFn.apply(LinearFn.apply(feat, self.fc.weight, self.fc.bias, 2, prec))   # tanh fused, bigru.py:18
