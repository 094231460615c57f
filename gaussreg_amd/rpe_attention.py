"""Mirror of geotransformer/modules/transformer/rpe_transformer.py:18-72 (RPEMultiHeadAttention), inference only.

The reference projects the (B,N,M,C) relative-position embedding through `proj_p` in every layer (77 GFLOP and a
602 MB temporary at N=M=767, C=256) before contracting it with q.  The contraction is linear in the embedding, so it
is re-associated here:  s_p[h,n,m] = emb[n,m,:] . (W_p[h]^T q[h,n,:]) + q[h,n,:] . b_p[h], and everything after the four
input projections -- q k^T, the positional term, scaling, factors / weights / masks, softmax and scores @ v -- runs in ONE
HIP kernel per batch element (gaussreg_amd/csrc/geo_embedding.hip: gr_rpe_attention): the embedding is streamed exactly
once per layer and no (H,N,M) or (N,M,C) intermediate goes through HBM.  The projections (nn.Linear) and the tiny
u = W_p^T q product stay torch ops.  State-dict keys are the reference's.
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
    def forward(self, input_q, input_k, input_v, embed_qk, key_weights=None, key_masks=None, attention_factors=None,
                lengths=None):
        """(B,N,C), (B,M,C), (B,M,C), (B,N,M,C) -> hidden_states (B,N,C), attention_scores (B,H,N,M).

        `lengths` (not in the reference; a list of B ints, self-attention only): the batch is a padded stack of B clouds of
        different sizes -- element b has lengths[b] real rows, `embed_qk` is then a LIST of B tensors (n_b, n_b, C).  The
        fused kernel runs per element with its true size, so the real rows are exactly what the unpadded call returns;
        padded rows of hidden_states are zero, attention_scores is None."""
        if lengths is not None:
            return self._forward_ragged(input_q, input_k, input_v, embed_qk, lengths)
        dev = _lib.require_gpu()
        L = _lib.lib()
        if not input_q.is_cuda:
            raise RuntimeError("RPEMultiHeadAttention: inputs must live on the GPU")
        dev = input_q.device
        B, N, C = input_q.shape
        M = input_k.shape[1]
        H, ch = self.num_heads, self.d_model_per_head
        q2 = self.proj_q(input_q).contiguous()                                   # (B,N,C), heads side by side
        k2 = self.proj_k(input_k).contiguous()
        v2 = self.proj_v(input_v).contiguous()
        wp = self.proj_p.weight.view(H, ch, C)                                  # rows h*ch..: head h
        qh = q2.view(B, N, H, ch)
        u = torch.einsum('bnhc,hcj->bnhj', qh, wp).contiguous()                 # (B,N,H,C)
        add = torch.einsum('bnhc,hc->bnh', qh, self.proj_p.bias.view(H, ch)).contiguous()
        emb = embed_qk.to(torch.float32).contiguous()
        scores = torch.empty((B, H, N, M), dtype=torch.float32, device=dev)
        hidden = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        fac = None if attention_factors is None else attention_factors.to(torch.float32).contiguous()
        kw = None if key_weights is None else key_weights.to(torch.float32).contiguous()
        km = None if key_masks is None else key_masks.to(torch.uint8).contiguous()
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            for b in range(B):
                _lib.check(L.gr_rpe_attention(_lib.ptr(emb[b]), _lib.ptr(u[b]), _lib.ptr(add[b]), _lib.ptr(q2[b]),
                                              _lib.ptr(k2[b]), _lib.ptr(v2[b]), _lib.ptr(None if fac is None else fac[b]),
                                              _lib.ptr(None if kw is None else kw[b]), _lib.ptr(None if km is None else km[b]),
                                              N, M, C, H, _lib.ptr(scores[b]), _lib.ptr(hidden[b]), st))
        if not isinstance(self.dropout, nn.Identity):
            scores = self.dropout(scores)  # inference: identity (the reference applies dropout to the scores before @ v)
        return hidden, scores


    @torch.no_grad()
    def _forward_ragged(self, input_q, input_k, input_v, embed_list, lengths):
        L = _lib.lib()
        dev = input_q.device
        B, N, C = input_q.shape
        H, ch = self.num_heads, self.d_model_per_head
        if input_k.shape != input_q.shape or len(embed_list) != B or len(lengths) != B:
            raise ValueError("lengths: self-attention over a padded stack, one embedding per element")
        q2 = self.proj_q(input_q).contiguous()
        k2 = self.proj_k(input_k).contiguous()
        v2 = self.proj_v(input_v).contiguous()
        wp = self.proj_p.weight.view(H, ch, C)
        qh = q2.view(B, N, H, ch)
        u = torch.einsum('bnhc,hcj->bnhj', qh, wp).contiguous()
        add = torch.einsum('bnhc,hc->bnh', qh, self.proj_p.bias.view(H, ch)).contiguous()
        hidden = torch.zeros((B, N, C), dtype=torch.float32, device=dev)
        nmax = max(int(n) for n in lengths)
        scores = torch.empty((H, nmax, nmax), dtype=torch.float32, device=dev)   # scratch: the kernel writes it, nobody reads
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            for b in range(B):
                n = int(lengths[b])
                emb = embed_list[b]
                if emb.shape != (n, n, C) or not emb.is_contiguous() or emb.dtype != torch.float32:
                    raise ValueError("embedding %d must be a contiguous float32 (n, n, C) tensor" % b)
                if n == 0:
                    continue
                _lib.check(L.gr_rpe_attention(_lib.ptr(emb), _lib.ptr(u[b]), _lib.ptr(add[b]), _lib.ptr(q2[b]), _lib.ptr(k2[b]),
                                              _lib.ptr(v2[b]), None, None, None, n, n, C, H, _lib.ptr(scores),
                                              _lib.ptr(hidden[b]), st))
        return hidden, None
