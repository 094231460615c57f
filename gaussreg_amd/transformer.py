"""Mirrors of the transformer stack GaussReg's coarse matcher runs on the superpoints, inference only:

  AttentionOutput            geotransformer/modules/transformer/output_layer.py:6-21
  MultiHeadAttention         geotransformer/modules/transformer/vanilla_transformer.py:15-69
  AttentionLayer             vanilla_transformer.py:72-103
  TransformerLayer           vanilla_transformer.py:106-132
  RPEAttentionLayer          geotransformer/modules/transformer/rpe_transformer.py:75-104
  RPETransformerLayer        rpe_transformer.py:107-132
  RPEConditionalTransformer  geotransformer/modules/transformer/conditional_transformer.py:73-117
  GeometricTransformer       geotransformer/modules/geotransformer/geotransformer.py:76-155

Constructors, forward signatures, return values and state-dict keys are the reference's, and sub-modules are created in
the reference's order (so a module built under the same torch.manual_seed carries the same weights).  The two kernels
that carry the cost are HIP: the structure embedding (gaussreg_amd.embedding, one fused kernel per cloud) and the
self-attention with the relative-position term (gaussreg_amd.rpe_attention, one fused kernel per batch element).  The
cross-attention has no (N,M,C) stream -- 767 x 767 x 256 is three small GEMMs and a softmax -- and stays on rocBLAS
through torch, like the nn.Linear / nn.LayerNorm glue around both.
"""
import torch
import torch.nn as nn

from . import _lib
from .embedding import GeometricStructureEmbedding
from .rpe_attention import RPEMultiHeadAttention

_ACTIVATIONS = {'ReLU': nn.ReLU, 'LeakyReLU': nn.LeakyReLU, 'ELU': nn.ELU, 'GELU': nn.GELU, 'Sigmoid': nn.Sigmoid,
                'Softplus': nn.Softplus, 'Tanh': nn.Tanh, 'Identity': nn.Identity}


def _activation(cfg):
    """layers/factory.py:71-80: a name or {'type': name, **kwargs}; LeakyReLU defaults to slope 0.2."""
    if cfg is None:
        return nn.Identity()
    kwargs = {} if isinstance(cfg, str) else {k: v for k, v in cfg.items() if k != 'type'}
    name = cfg if isinstance(cfg, str) else cfg['type']
    if name not in _ACTIVATIONS:
        raise AssertionError(f'Illegal activation: {name}.')
    if name == 'LeakyReLU':
        kwargs.setdefault('negative_slope', 0.2)
    return _ACTIVATIONS[name](**kwargs)


def _dropout(p):
    return nn.Identity() if p is None or p == 0 else nn.Dropout(p=p)


class AttentionOutput(nn.Module):
    def __init__(self, d_model, dropout=None, activation_fn='ReLU'):
        super().__init__()
        self.expand = nn.Linear(d_model, d_model * 2)
        self.activation = _activation(activation_fn)
        self.squeeze = nn.Linear(d_model * 2, d_model)
        self.dropout = _dropout(dropout)
        self.norm = nn.LayerNorm(d_model)

    @torch.no_grad()
    def forward(self, input_states):
        grown = self.squeeze(self.activation(self.expand(input_states)))
        return self.norm(input_states + self.dropout(grown))


class MultiHeadAttention(nn.Module):
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
        self.dropout = _dropout(dropout)

    @torch.no_grad()
    def forward(self, input_q, input_k, input_v, key_weights=None, key_masks=None, attention_factors=None,
                attention_masks=None):
        """(B,N,C), (B,M,C), (B,M,C) -> hidden_states (B,N,C), attention_scores (B,H,N,M)."""
        _lib.require_gpu()
        B, N, _ = input_q.shape
        M, H, ch = input_k.shape[1], self.num_heads, self.d_model_per_head
        q = self.proj_q(input_q).view(B, N, H, ch).transpose(1, 2)
        k = self.proj_k(input_k).view(B, M, H, ch).transpose(1, 2)
        v = self.proj_v(input_v).view(B, M, H, ch).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(2, 3)) / ch ** 0.5
        if attention_factors is not None:
            scores = attention_factors.unsqueeze(1) * scores
        if key_weights is not None:
            scores = scores * key_weights[:, None, None, :]
        if key_masks is not None:
            scores = scores.masked_fill(key_masks[:, None, None, :], float('-inf'))
        if attention_masks is not None:
            scores = scores.masked_fill(attention_masks, float('-inf'))
        scores = self.dropout(torch.softmax(scores, dim=-1))
        hidden = torch.matmul(scores, v).transpose(1, 2).reshape(B, N, H * ch)
        return hidden, scores


class AttentionLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__()
        self.attention = MultiHeadAttention(d_model, num_heads, dropout=dropout)
        self.linear = nn.Linear(d_model, d_model)
        self.dropout = _dropout(dropout)
        self.norm = nn.LayerNorm(d_model)

    @torch.no_grad()
    def forward(self, input_states, memory_states, memory_weights=None, memory_masks=None, attention_factors=None,
                attention_masks=None):
        hidden, scores = self.attention(input_states, memory_states, memory_states, key_weights=memory_weights,
                                        key_masks=memory_masks, attention_factors=attention_factors,
                                        attention_masks=attention_masks)
        return self.norm(self.dropout(self.linear(hidden)) + input_states), scores


class TransformerLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU'):
        super().__init__()
        self.attention = AttentionLayer(d_model, num_heads, dropout=dropout)
        self.output = AttentionOutput(d_model, dropout=dropout, activation_fn=activation_fn)

    @torch.no_grad()
    def forward(self, input_states, memory_states, memory_weights=None, memory_masks=None, attention_factors=None,
                attention_masks=None):
        hidden, scores = self.attention(input_states, memory_states, memory_weights=memory_weights,
                                        memory_masks=memory_masks, attention_factors=attention_factors,
                                        attention_masks=attention_masks)
        return self.output(hidden), scores


class RPEAttentionLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__()
        self.attention = RPEMultiHeadAttention(d_model, num_heads, dropout=dropout)
        self.linear = nn.Linear(d_model, d_model)
        self.dropout = _dropout(dropout)
        self.norm = nn.LayerNorm(d_model)

    @torch.no_grad()
    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None,
                attention_factors=None, lengths=None):
        hidden, scores = self.attention(input_states, memory_states, memory_states, position_states,
                                        key_weights=memory_weights, key_masks=memory_masks,
                                        attention_factors=attention_factors, lengths=lengths)
        return self.norm(self.dropout(self.linear(hidden)) + input_states), scores


class RPETransformerLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU'):
        super().__init__()
        self.attention = RPEAttentionLayer(d_model, num_heads, dropout=dropout)
        self.output = AttentionOutput(d_model, dropout=dropout, activation_fn=activation_fn)

    @torch.no_grad()
    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None,
                attention_factors=None, lengths=None):
        hidden, scores = self.attention(input_states, memory_states, position_states, memory_weights=memory_weights,
                                        memory_masks=memory_masks, attention_factors=attention_factors, lengths=lengths)
        return self.output(hidden), scores


class RPEConditionalTransformer(nn.Module):
    def __init__(self, blocks, d_model, num_heads, dropout=None, activation_fn='ReLU', return_attention_scores=False,
                 parallel=False):
        super().__init__()
        self.blocks = blocks
        layers = []
        for block in self.blocks:
            if block not in ('self', 'cross'):
                raise ValueError('Unsupported block type "{}".'.format(block))
            kind = RPETransformerLayer if block == 'self' else TransformerLayer
            layers.append(kind(d_model, num_heads, dropout=dropout, activation_fn=activation_fn))
        self.layers = nn.ModuleList(layers)
        self.return_attention_scores = return_attention_scores
        self.parallel = parallel

    @torch.no_grad()
    def forward(self, feats0, feats1, embeddings0, embeddings1, masks0=None, masks1=None, lengths0=None, lengths1=None):
        """lengths0 / lengths1 (not in the reference): a padded stack of clouds of different sizes -- the self blocks then
        take one (n_b, n_b, C) embedding per element (lists) and run every element at its true size (RPEMultiHeadAttention);
        the cross blocks use masks0 / masks1 (True = padding) as the reference does."""
        kept = []
        for layer, block in zip(self.layers, self.blocks):
            if block == 'self':
                feats0, scores0 = layer(feats0, feats0, embeddings0, memory_masks=masks0, lengths=lengths0)
                feats1, scores1 = layer(feats1, feats1, embeddings1, memory_masks=masks1, lengths=lengths1)
            elif self.parallel:
                # both directions read the features of the previous layer
                out0, scores0 = layer(feats0, feats1, memory_masks=masks1)
                out1, scores1 = layer(feats1, feats0, memory_masks=masks0)
                feats0, feats1 = out0, out1
            else:
                # sequential (the reference's default): the second direction already sees the updated feats0
                feats0, scores0 = layer(feats0, feats1, memory_masks=masks1)
                feats1, scores1 = layer(feats1, feats0, memory_masks=masks0)
            if self.return_attention_scores:
                kept.append([scores0, scores1])
        if self.return_attention_scores:
            return feats0, feats1, kept
        return feats0, feats1


class GeometricTransformer(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_heads, blocks, sigma_d, sigma_a, angle_k, dropout=None,
                 activation_fn='ReLU', reduction_a='max'):
        super().__init__()
        self.embedding = GeometricStructureEmbedding(hidden_dim, sigma_d, sigma_a, angle_k, reduction_a=reduction_a)
        self.in_proj = nn.Linear(input_dim, hidden_dim)
        self.transformer = RPEConditionalTransformer(blocks, hidden_dim, num_heads, dropout=dropout,
                                                     activation_fn=activation_fn)
        self.out_proj = nn.Linear(hidden_dim, output_dim)

    @torch.no_grad()
    def forward(self, ref_points, src_points, ref_feats, src_feats, ref_masks=None, src_masks=None, ref_lengths=None,
                src_lengths=None):
        """(B,N,3), (B,M,3), (B,N,Cin), (B,M,Cin) -> (B,N,Cout), (B,M,Cout).

        ref_lengths / src_lengths (not in the reference; lists of B ints, together with ref_masks / src_masks = True on the
        padding): several scene pairs of different sizes as one padded batch -- the structure embedding and the
        self-attention run per cloud at its true size (no padded point can become a neighbour or a key), the cross-attention
        masks the padding; the rows of the real superpoints are those of the pair alone up to GEMM summation order."""
        if ref_lengths is not None:
            ref_embeddings = [self.embedding(ref_points[b:b + 1, :n])[0] for b, n in enumerate(ref_lengths)]
            src_embeddings = [self.embedding(src_points[b:b + 1, :n])[0] for b, n in enumerate(src_lengths)]
        else:
            ref_embeddings = self.embedding(ref_points)
            src_embeddings = self.embedding(src_points)
        ref_feats, src_feats = self.transformer(self.in_proj(ref_feats), self.in_proj(src_feats), ref_embeddings,
                                                src_embeddings, masks0=ref_masks, masks1=src_masks, lengths0=ref_lengths,
                                                lengths1=src_lengths)
        return self.out_proj(ref_feats), self.out_proj(src_feats)
