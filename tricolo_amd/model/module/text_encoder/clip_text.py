"""CLIPTextEncoder - drop-in for /root/reference/tricolo/model/module/text_encoder/clip_text.py:6-22.

CLIP itself never runs on the hot path: the reference reads the cached 768-d vector
data_dict["clip_embeddings_text"] and applies Linear -> ReLU -> Dropout(0.1) -> Linear (no normalise).  `clip_model`
is only consulted for `.visual.output_dim`; without one (CLIP weights cannot be downloaded here) `clip_dim` is used.
Like the reference, a batch without the cached key is an error.
"""
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ....layers import LinearFn, TriModule, require_gpu


class CLIPTextEncoder(TriModule):
    def __init__(self, out_dim, clip_model=None, clip_dim=768, precision=None, **kwargs):
        super().__init__()
        in_dim = clip_model.visual.output_dim if clip_model is not None else clip_dim
        self.mlp = nn.Sequential(nn.Linear(in_dim, out_dim), nn.ReLU(inplace=True), nn.Dropout(0.1), nn.Linear(out_dim, out_dim))
        self.precision = precision

    def forward(self, tokens, data_dict):
        if "clip_embeddings_text" not in data_dict:
            raise UnboundLocalError("clip_embeddings_text missing from the batch (clip_text.py:17-21 has no live encode_text path)")
        x = data_dict["clip_embeddings_text"]
        require_gpu(x, "CLIPTextEncoder")
        prec = self.precision or ops.default_precision()
        h = LinearFn.apply(x, self.mlp[0].weight, self.mlp[0].bias, 1, prec)
        h = F.dropout(h, self.mlp[2].p, self.mlp[2].training)
        return LinearFn.apply(h, self.mlp[3].weight, self.mlp[3].bias, 0, prec)
