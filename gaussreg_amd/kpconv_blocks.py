"""Mirror of geotransformer/modules/kpconv/modules.py (inference): the blocks KPConvFPN is assembled from
(backbone.py:95-162), with the HIP KPConv / maxpool / nearest_upsample / GroupNorm inside.  Sub-module names and parameter
shapes follow the reference so its checkpoints load key for key (`KPConv.weights`, `norm.norm.weight`, `mlp.weight`, ...).
nn.Linear stays the stock PyTorch-ROCm layer (rocBLAS), as in the reference."""
import torch
import torch.nn as nn

from . import _lib
from .kpconv import KPConv, maxpool, nearest_upsample


import contextlib
import threading

_segments = threading.local()


@contextlib.contextmanager
def norm_segments(table):
    """Several stack-mode batches (scene pairs) through the backbone in ONE pass: `table` maps the row count of a pyramid
    level to (device int64 offsets of the pairs' row ranges at that level, rows of the longest range).  Inside the context
    every GroupNorm whose input has such a row count normalises each range with its own statistics -- what the reference
    computes when every pair goes through the network alone (modules.py:32-50: the statistics run over ALL points of the
    collated batch, i.e. of one pair)."""
    old = getattr(_segments, "table", None)
    for n, (seg_off, _) in table.items():
        if seg_off.dim() != 1 or seg_off.numel() < 2:
            raise ValueError("norm_segments: offsets must be a 1-D tensor of B + 1 entries")
    _segments.table = table
    try:
        yield
    finally:
        _segments.table = old


class GroupNorm(nn.Module):
    """modules.py:32-50: GroupNorm over the channel axis of a (N, C) point-feature matrix."""

    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_groups, self.num_channels = num_groups, num_channels
        self.norm = nn.GroupNorm(num_groups, num_channels)

    def forward(self, x, negative_slope=None, residual=None):
        """`negative_slope`: fuse the LeakyReLU that follows the norm in every block (None: plain GroupNorm).
        `residual` (same shape as x): added between the norm and the activation -- the tail of a residual block."""
        c = self.num_channels
        hip_ok = (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not torch.is_grad_enabled() and c % 4 == 0
                  and ((c // 4 <= 256 and 256 % (c // 4) == 0) or (c // 4) % 256 == 0) and self.num_groups <= 64)
        if not hip_ok:
            y = self.norm(x.t().unsqueeze(0))     # (N, C) -> (1, C, N): statistics per group over all points
            y = y.squeeze(0).t().squeeze()        # the trailing squeeze() is the reference's (modules.py:50)
            if residual is not None:
                y = y + residual
            return y if negative_slope is None else nn.functional.leaky_relu(y, negative_slope)
        L = _lib.lib()
        x = x.contiguous()
        out = torch.empty_like(x)
        dev = x.device
        table = getattr(_segments, "table", None)
        seg = table.get(x.shape[0]) if table else None
        with torch.cuda.device(dev):
            w, b = self.norm.weight, self.norm.bias
            if residual is not None:
                res = residual.reshape(x.shape).contiguous()
                seg_off, max_rows = seg if seg is not None else (None, 0)
                nseg = seg_off.numel() - 1 if seg is not None else 1
                ws = _lib.workspace(dev, L.gr_group_norm_seg_workspace_bytes(self.num_groups, nseg))
                _lib.check(L.gr_group_norm_res(_lib.ptr(x), x.shape[0], c, self.num_groups,
                                               _lib.ptr(None if w is None else w.detach().contiguous()),
                                               _lib.ptr(None if b is None else b.detach().contiguous()), float(self.norm.eps),
                                               1.0 if negative_slope is None else float(negative_slope), _lib.ptr(res),
                                               _lib.ptr(out), _lib.ptr(seg_off), nseg, int(max_rows), _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr(dev)))
                # the reference squeezes the norm's output BEFORE the shortcut is added (modules.py:50, :217-222): for a
                # single row the sum broadcasts back to the shortcut's (1, C)
                return out.reshape(torch.broadcast_shapes(out.squeeze().shape, residual.shape))
            if seg is not None:
                seg_off, max_rows = seg
                nseg = seg_off.numel() - 1
                ws = _lib.workspace(dev, L.gr_group_norm_seg_workspace_bytes(self.num_groups, nseg))
                _lib.check(L.gr_group_norm_seg(_lib.ptr(x), x.shape[0], c, self.num_groups,
                                               _lib.ptr(None if w is None else w.detach().contiguous()),
                                               _lib.ptr(None if b is None else b.detach().contiguous()), float(self.norm.eps),
                                               1.0 if negative_slope is None else float(negative_slope), _lib.ptr(out),
                                               _lib.ptr(seg_off), nseg, int(max_rows), _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr(dev)))
                return out.squeeze()
            ws = _lib.workspace(dev, L.gr_group_norm_workspace_bytes(self.num_groups))
            _lib.check(L.gr_group_norm(_lib.ptr(x), x.shape[0], c, self.num_groups,
                                       _lib.ptr(None if w is None else w.detach().contiguous()),
                                       _lib.ptr(None if b is None else b.detach().contiguous()), float(self.norm.eps),
                                       1.0 if negative_slope is None else float(negative_slope), _lib.ptr(out), _lib.ptr(ws),
                                       ws.numel(), _lib.stream_ptr(dev)))
        return out.squeeze()


def segment_table(level_lengths, device):
    """The table norm_segments takes, from the pyramid's per-level `lengths` (clouds stacked pair by pair: cloud 2 b = ref,
    2 b + 1 = src of pair b).  Two levels with the same total row count would be indistinguishable by row count (the key the
    GroupNorm layers look their segments up by) -- that raises here instead of normalising with the wrong offsets."""
    table = {}
    for lengths in level_lengths:
        ll = [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
        offs = [0]
        for b in range(len(ll) // 2):
            offs.append(offs[-1] + ll[2 * b] + ll[2 * b + 1])
        entry = (torch.tensor(offs, dtype=torch.int64, device=device), max(offs[i + 1] - offs[i] for i in range(len(offs) - 1)))
        if offs[-1] in table and table[offs[-1]][0].tolist() != offs:
            raise ValueError(f"norm_segments: two pyramid levels have {offs[-1]} rows in total but different pair boundaries")
        table[offs[-1]] = entry
    return table


def _norm(out_channels, group_norm, layer_norm):
    return nn.LayerNorm(out_channels) if layer_norm else GroupNorm(group_norm, out_channels)


def _norm_act(norm, x, negative_slope):
    """norm -> LeakyReLU, in one kernel when the norm is the HIP GroupNorm."""
    if isinstance(norm, GroupNorm):
        return norm(x, negative_slope)
    return nn.functional.leaky_relu(norm(x), negative_slope)


class UnaryBlock(nn.Module):
    """modules.py:53-83: Linear -> norm -> (LeakyReLU 0.1)."""

    def __init__(self, in_channels, out_channels, group_norm, has_relu=True, bias=True, layer_norm=False):
        super().__init__()
        self.in_channels, self.out_channels, self.group_norm = in_channels, out_channels, group_norm
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)
        self.norm = _norm(out_channels, group_norm, layer_norm)
        self.leaky_relu = nn.LeakyReLU(0.1) if has_relu else None

    def forward(self, x):
        x = self.mlp(x)
        return self.norm(x) if self.leaky_relu is None else _norm_act(self.norm, x, self.leaky_relu.negative_slope)


class LastUnaryBlock(nn.Module):
    """modules.py:86-101: a bare Linear."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x):
        return self.mlp(x)


class ConvBlock(nn.Module):
    """modules.py:104-145: KPConv -> norm -> LeakyReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, group_norm, negative_slope=0.1, bias=True,
                 layer_norm=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.KPConv = KPConv(in_channels, out_channels, kernel_size, radius, sigma, bias=bias)
        self.norm = _norm(out_channels, group_norm, layer_norm)
        self.leaky_relu = nn.LeakyReLU(negative_slope=negative_slope)

    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        return _norm_act(self.norm, self.KPConv(s_feats, q_points, s_points, neighbor_indices), self.leaky_relu.negative_slope)


class ResidualBlock(nn.Module):
    """modules.py:148-225: bottleneck -- unary1 (C_in -> C_out/4), KPConv on C_out/4 channels, unary2 (-> C_out), plus a
    shortcut that is max-pooled over the neighbourhood when the block is strided."""

    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, group_norm, strided=False, bias=True,
                 layer_norm=False):
        super().__init__()
        self.in_channels, self.out_channels, self.strided = in_channels, out_channels, strided
        mid = out_channels // 4
        self.unary1 = (UnaryBlock(in_channels, mid, group_norm, bias=bias, layer_norm=layer_norm)
                       if in_channels != mid else nn.Identity())
        self.KPConv = KPConv(mid, mid, kernel_size, radius, sigma, bias=bias)
        self.norm_conv = _norm(mid, group_norm, layer_norm)
        self.unary2 = UnaryBlock(mid, out_channels, group_norm, has_relu=False, bias=bias, layer_norm=layer_norm)
        self.unary_shortcut = (UnaryBlock(in_channels, out_channels, group_norm, has_relu=False, bias=bias,
                                          layer_norm=layer_norm) if in_channels != out_channels else nn.Identity())
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        x = self.KPConv(self.unary1(s_feats), q_points, s_points, neighbor_indices)
        h = _norm_act(self.norm_conv, x, self.leaky_relu.negative_slope)
        shortcut = maxpool(s_feats, neighbor_indices) if self.strided else s_feats
        sc = self.unary_shortcut(shortcut)
        if isinstance(self.unary2.norm, GroupNorm) and not torch.is_grad_enabled():
            # unary2's GroupNorm (no activation), + shortcut, LeakyReLU: one apply pass instead of three
            y = self.unary2.mlp(h)
            if sc.shape == y.shape:
                return self.unary2.norm(y, self.leaky_relu.negative_slope, residual=sc)
        return self.leaky_relu(self.unary2(h) + sc)


class KPConvFPN(nn.Module):
    """experiments/geotransformer.gaussian_splatting.indoor/backbone.py:95-212: 5-stage encoder, 3-stage decoder."""

    def __init__(self, input_dim, output_dim, init_dim, kernel_size, init_radius, init_sigma, group_norm):
        super().__init__()
        C, K, r, s, g = init_dim, kernel_size, init_radius, init_sigma, group_norm
        self.encoder1_1 = ConvBlock(input_dim, C, K, r, s, g)
        self.encoder1_2 = ResidualBlock(C, C * 2, K, r, s, g)
        self.encoder2_1 = ResidualBlock(C * 2, C * 2, K, r, s, g, strided=True)
        self.encoder2_2 = ResidualBlock(C * 2, C * 4, K, r * 2, s * 2, g)
        self.encoder2_3 = ResidualBlock(C * 4, C * 4, K, r * 2, s * 2, g)
        self.encoder3_1 = ResidualBlock(C * 4, C * 4, K, r * 2, s * 2, g, strided=True)
        self.encoder3_2 = ResidualBlock(C * 4, C * 8, K, r * 4, s * 4, g)
        self.encoder3_3 = ResidualBlock(C * 8, C * 8, K, r * 4, s * 4, g)
        self.encoder4_1 = ResidualBlock(C * 8, C * 8, K, r * 4, s * 4, g, strided=True)
        self.encoder4_2 = ResidualBlock(C * 8, C * 16, K, r * 8, s * 8, g)
        self.encoder4_3 = ResidualBlock(C * 16, C * 16, K, r * 8, s * 8, g)
        self.encoder5_1 = ResidualBlock(C * 16, C * 16, K, r * 8, s * 8, g, strided=True)
        self.encoder5_2 = ResidualBlock(C * 16, C * 32, K, r * 16, s * 16, g)
        self.encoder5_3 = ResidualBlock(C * 32, C * 32, K, r * 16, s * 16, g)
        self.decoder4 = UnaryBlock(C * 48, C * 16, g)
        self.decoder3 = UnaryBlock(C * 24, C * 8, g)
        self.decoder2 = LastUnaryBlock(C * 12, output_dim)

    @torch.no_grad()
    def forward(self, feats, data_dict):
        pts, nb, sub, up = data_dict['points'], data_dict['neighbors'], data_dict['subsampling'], data_dict['upsampling']
        feats_list = []
        f1 = self.encoder1_2(self.encoder1_1(feats, pts[0], pts[0], nb[0]), pts[0], pts[0], nb[0])
        f2 = self.encoder2_1(f1, pts[1], pts[0], sub[0])
        f2 = self.encoder2_3(self.encoder2_2(f2, pts[1], pts[1], nb[1]), pts[1], pts[1], nb[1])
        f3 = self.encoder3_1(f2, pts[2], pts[1], sub[1])
        f3 = self.encoder3_3(self.encoder3_2(f3, pts[2], pts[2], nb[2]), pts[2], pts[2], nb[2])
        f4 = self.encoder4_1(f3, pts[3], pts[2], sub[2])
        f4 = self.encoder4_3(self.encoder4_2(f4, pts[3], pts[3], nb[3]), pts[3], pts[3], nb[3])
        f5 = self.encoder5_1(f4, pts[4], pts[3], sub[3])
        f5 = self.encoder5_3(self.encoder5_2(f5, pts[4], pts[4], nb[4]), pts[4], pts[4], nb[4])
        feats_list.append(f5)
        l4 = self.decoder4(torch.cat([nearest_upsample(f5, up[3]), f4], dim=1))
        feats_list.append(l4)
        l3 = self.decoder3(torch.cat([nearest_upsample(l4, up[2]), f3], dim=1))
        feats_list.append(l3)
        l2 = self.decoder2(torch.cat([nearest_upsample(l3, up[1]), f2], dim=1))
        feats_list.append(l2)
        feats_list.reverse()
        return feats_list
