"""Mirror of geotransformer/modules/transformer/rpe_transformer.py:18-72 (RPEMultiHeadAttention), inference only.

The reference projects the (B,N,M,C) relative-position embedding through `proj_p` in every layer (77 GFLOP and a
602 MB temporary at N=M=767, C=256) before contracting it with q.  The contraction is linear in the embedding, so it
is re-associated here:  s_p[h,n,m] = emb[n,m,:] . (W_p[h]^T q[h,n,:]) + q[h,n,:] . b_p[h]  -- one memory-bound pass over
the embedding in a HIP kernel (gaussreg_amd/csrc/geo_embedding.hip: gr_rpe_scores).  The small dense products
(q/k/v projections, q k^T, softmax, scores @ v) stay plain torch ops (rocBLAS).  State-dict keys are the reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


class RPEMultiHeadAttention(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        self.d_model = d_model
        self.num_heads = num_heads
        self.d_model_per_head = d_model // num_heads
        self.proj_q = nn.Linear(self.d_model, self.d_model)
        self.proj_k = nn.Linear(self.d_model, self.d_model)
        self.proj_v = nn.Linear(self.d_model, self.d_model)
        self.proj_p = nn.Linear(self.d_model, self.d_model)
        self.dropout = nn.Identity() if dropout is None or dropout <= 0 else nn.Dropout(dropout)

    @torch.no_grad()
    def forward(self, input_q, input_k, input_v, embed_qk, key_weights=None, key_masks=None, attention_factors=None):
        """(B,N,C), (B,M,C), (B,M,C), (B,N,M,C) -> hidden_states (B,N,C), attention_scores (B,H,N,M)."""
        dev = _lib.require_gpu()
        L = _lib.lib()
        if not input_q.is_cuda:
            raise RuntimeError("RPEMultiHeadAttention: inputs must live on the GPU")
        dev = input_q.device
        B, N, C = input_q.shape
        M = input_k.shape[1]
        H, ch = self.num_heads, self.d_model_per_head
        q = self.proj_q(input_q).view(B, N, H, ch).permute(0, 2, 1, 3)          # (B,H,N,c)
        k = self.proj_k(input_k).view(B, M, H, ch).permute(0, 2, 1, 3)
        v = self.proj_v(input_v).view(B, M, H, ch).permute(0, 2, 1, 3)
        wp = self.proj_p.weight.view(H, ch, C)                                  # rows h*ch..: head h
        u = torch.einsum('bhnc,hcj->bnhj', q, wp).contiguous()                  # (B,N,H,C)
        add = torch.einsum('bhnc,hc->bnh', q, self.proj_p.bias.view(H, ch)).contiguous()
        emb = embed_qk.to(torch.float32).contiguous()
        scores_p = torch.empty((B, H, N, M), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            for b in range(B):
                _lib.check(L.gr_rpe_scores(_lib.ptr(emb[b]), _lib.ptr(u[b]), _lib.ptr(add[b]), N, M, C, H,
                                           _lib.ptr(scores_p[b]), _lib.stream_ptr(dev)))
        scores = (torch.matmul(q, k.transpose(-1, -2)) + scores_p) / ch ** 0.5
        if attention_factors is not None:
            scores = attention_factors.unsqueeze(1) * scores
        if key_weights is not None:
            scores = scores * key_weights.unsqueeze(1).unsqueeze(1)
        if key_masks is not None:
            scores = scores.masked_fill(key_masks.unsqueeze(1).unsqueeze(1), float('-inf'))
        scores = self.dropout(F.softmax(scores, dim=-1))
        hidden = torch.matmul(scores, v).permute(0, 2, 1, 3).reshape(B, N, C)
        return hidden, scores
