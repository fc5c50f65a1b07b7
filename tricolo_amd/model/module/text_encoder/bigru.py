"""BiGRUEncoder on hand-written gfx950 kernels - drop-in for
/root/reference/tricolo/model/module/text_encoder/bigru.py:8-18.

Same constructor (vocab_size, out_dim, **kwargs - `clip_model` ignored), same forward(tokens [B,L] int, data_dict)
-> [B, out_dim] unit rows, same state-dict keys (embedding_layer.weight, gru.{weight_ih,weight_hh,bias_ih,bias_hh}_l0
[_reverse], fc.{weight,bias}; `gru` is a real nn.GRU used only as the parameter holder).  As in the reference there is
no packing: pad tokens (id 0, zero embedding row) are stepped through the GRU and the forward final state is taken
after the trailing pads.

MI355X design: the embedding gather stays a torch op (one launch each way); the input projection of all 96 steps and
both directions is ONE MFMA GEMM ([L*B,256] x [256,768], bias fused); the recurrence is ONE persistent kernel per
direction-batch-chunk with W_hh resident in registers as MFMA fragments (tricolo_amd/csrc/gru.hip) instead of the
~3,900 MIOpen launches per training step measured for nn.GRU on ROCm (profiles/r1); the weight gradients are three
GEMMs over the stored gate gradients; fc + tanh is the dense MFMA kernel and the normalise is the row kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ....layers import L2NormFn, LinearFn, TriModule, linear_fwd, linear_geom, require_gpu


class _BiGRUFn(torch.autograd.Function):
    """emb [L,B,256] + the eight nn.GRU parameters -> cat(h_fwd_final, h_rev_final) [B,256]."""

    @staticmethod
    def forward(ctx, emb, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, precision):
        L, B, I = emb.shape
        x2d = emb.contiguous().view(L * B, I)
        # [w_ih_f; w_ih_r], [b_ih_f; b_ih_r], stack(w_hh), stack(b_hh) in one buffer, ONE copy launch (the eight nn.GRU
        # parameters keep their reference names / shapes; the fused kernels want the two directions side by side)
        buf = torch.empty((768 * I + 768 + 768 * 128 + 768,), dtype=torch.float32, device=emb.device)
        w_ih, b_ih = buf[:768 * I].view(768, I), buf[768 * I:768 * I + 768]
        w_hh = buf[768 * I + 768:768 * I + 768 + 768 * 128].view(2, 384, 128)
        b_hh = buf[768 * I + 768 + 768 * 128:].view(2, 384)
        ops.copy_segments([(w_ih_f.contiguous(), w_ih[:384]), (w_ih_r.contiguous(), w_ih[384:]), (b_ih_f, b_ih[:384]), (b_ih_r, b_ih[384:]),
                           (w_hh_f.contiguous(), w_hh[0]), (w_hh_r.contiguous(), w_hh[1]), (b_hh_f, b_hh[0]), (b_hh_r, b_hh[1])])
        fine = ops.TIMELINE is not None and ops.TIMELINE.get("fine")
        xproj = linear_fwd(x2d, w_ih, b_ih, 0, precision)               # [L*B, 768]
        if fine:
            ops.stamp("text.fwd.xproj.end")
        hfinal, hs, gates = ops.gru_fwd(xproj, w_hh, b_hh, B, L, precision)
        if fine:
            ops.stamp("text.fwd.gru.end")
        ctx.save_for_backward(x2d, w_ih, w_hh, hs, gates)
        ctx.dims, ctx.precision = (L, B, I), precision
        return hfinal

    @staticmethod
    def backward(ctx, dhfinal):
        x2d, w_ih, w_hh, hs, gates = ctx.saved_tensors
        (L, B, I), prec = ctx.dims, ctx.precision
        ops.stamp("text.bwd.gru.start")
        dgi, dgh, hprev, dbias = ops.gru_bwd(dhfinal, w_hh, hs, gates, B, L, prec)
        ops.stamp("text.bwd.gru.end")
        # dx first (the embedding gradient waits for it), then the three weight gradients with ONE grouped reduce behind them (round 5:
        # this tail ends the Bi(V) step's backward; a side stream for the weight gradients would have to be joined back into this
        # tower's own side stream, and a join into a stream that is itself a forked branch crashes hipStreamEndCapture on this ROCm -
        # tools/probes/graph_nested_fork.py)
        hp = ops.head_precision(prec)
        g_ih, g_hh = linear_geom(L * B, I, 768), linear_geom(L * B, 128, 384)
        dgi = dgi.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_dgrad(dgi, g_ih, ops.pack_weight(w_ih, g_ih, hp, transposed=True)).view(x2d.shape)
        batch = ops.wgrad_batch(dgi.device)
        dw_ih = ops.conv_wgrad(x2d, dgi, g_ih, w_ih, hp, batch=batch)
        dw_hh = [ops.conv_wgrad(hprev[d], dgh[d], g_hh, w_hh[d], prec, batch=batch) for d in range(2)]
        if batch is not None:
            batch.flush()
        db_ih_f, db_hh_f, db_ih_r, db_hh_r = ops.gru_bias_grads(dbias)     # per-chunk (dr, dz, dn_input, dn_hidden) sums -> nn.GRU biases
        if ops.DEBUG_KEEP is not None:                                     # (tools/r6/determinism.py: which tensor stops being reproducible)
            ops.DEBUG_KEEP.update(gru_dhfinal=dhfinal, gru_hs=hs, gru_gates=gates, gru_dgi=dgi, gru_dgh=dgh, gru_hprev=hprev, gru_dbias=dbias,
                                  gru_dx=dx, gru_dw_ih=dw_ih, gru_dw_hh0=dw_hh[0], gru_dw_hh1=dw_hh[1], gru_x2d=x2d)
        demb = dx.view(L, B, I) if dx is not None else None
        return (demb, dw_ih[:384], dw_hh[0], db_ih_f, db_hh_f, dw_ih[384:], dw_hh[1], db_ih_r, db_hh_r, None)


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding(padding_idx=0) lookup straight into time-major order, deterministic dense weight gradient."""

    @staticmethod
    def forward(ctx, tokens, weight):
        tok = tokens.to(torch.int32).contiguous()
        ctx.save_for_backward(tok)
        ctx.vocab = weight.shape[0]
        return ops.embedding_fwd(tok, weight)

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        dw = ops.embedding_bwd(tok, dout, ctx.vocab, padding_idx=0)
        if ops.TIMELINE is not None and ops.TIMELINE.get("fine"):
            ops.stamp("text.bwd.end")
        return None, dw


class BiGRUEncoder(TriModule):
    def __init__(self, vocab_size, out_dim, precision=None, **kwargs):
        super().__init__()
        self.embedding_layer = nn.Embedding(vocab_size, 256, padding_idx=0)
        self.gru = nn.GRU(input_size=256, hidden_size=128, num_layers=1, bidirectional=True)
        self.fc = nn.Linear(256, out_dim)
        self.precision = precision

    def forward(self, x, data_dict=None):
        require_gpu(x, "BiGRUEncoder")
        prec = self.precision or ops.default_precision()
        if ops.TIMELINE is not None and ops.TIMELINE.get("fine"):
            ops.stamp("text.fwd.start")
        emb = _EmbeddingFn.apply(x, self.embedding_layer.weight)                                        # [L,B,256], bigru.py:15
        g = self.gru
        feat = _BiGRUFn.apply(emb, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0, g.weight_ih_l0_reverse,
                              g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse, prec)   # bigru.py:16-17
        return L2NormFn.apply(LinearFn.apply(feat, self.fc.weight, self.fc.bias, 2, prec))              # bigru.py:18
