"""Mirror of geotransformer/modules/kpconv (forward, inference): KPConv + maxpool + nearest_upsample on HIP.

`KPConv` keeps the reference's parameters / buffers (`weights` (K,Cin,Cout), optional `bias`, buffer
`kernel_points` (K,3)) so reference checkpoints load with the same state-dict keys (kpconv.py:54-65).
The kernel-point generator of the reference (kernel_points.py, needs open3d + a PLY asset) is not
reproduced: kernel points come from the checkpoint or are passed in.
"""
import math

import torch
import torch.nn as nn

from . import _lib


def _f32(t, dev):
    return (t if t.is_cuda else t.to(dev)).to(torch.float32).contiguous()


class KPConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, bias=False, dimension=3, inf=1e6,
                 eps=1e-9, kernel_points=None):
        super().__init__()
        self.kernel_size, self.in_channels, self.out_channels = kernel_size, in_channels, out_channels
        self.radius, self.sigma, self.dimension, self.inf, self.eps = radius, sigma, dimension, inf, eps
        self.weights = nn.Parameter(torch.zeros(kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))  # kpconv.py:68
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weights)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)
        kp = torch.zeros(kernel_size, dimension) if kernel_points is None else torch.as_tensor(kernel_points).float()
        self.register_buffer('kernel_points', kp)

    @torch.no_grad()
    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        dev = _lib.require_gpu()
        L = _lib.lib()
        out_device = s_feats.device
        f = _f32(s_feats, dev)
        dev = f.device
        q, s = _f32(q_points, dev), _f32(s_points, dev)
        nb = neighbor_indices.to(device=dev, dtype=torch.int64).contiguous()
        kp = _f32(self.kernel_points, dev)
        wts = _f32(self.weights.detach(), dev)
        b = None if self.bias is None else _f32(self.bias.detach(), dev)
        N, Cin = f.shape
        M, H = nb.shape
        K, _, Cout = wts.shape
        out = torch.empty((M, Cout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_kpconv_workspace_bytes(N, M, K, Cin))
            _lib.check(L.gr_kpconv_forward(_lib.ptr(f), _lib.ptr(q), _lib.ptr(s), _lib.ptr(nb), N, M, H, Cin, Cout,
                                           _lib.ptr(kp), K, _lib.ptr(wts), _lib.ptr(b), float(self.sigma),
                                           float(self.inf), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(dev)))
        return out if out_device.type == "cuda" else out.to(out_device)


def _pool(x, neighbor_indices, mode):
    dev = _lib.require_gpu()
    L = _lib.lib()
    out_device = x.device
    xx = _f32(x, dev)
    dev = xx.device
    nb = neighbor_indices.to(device=dev, dtype=torch.int64).contiguous()
    N, C = xx.shape
    M, H = nb.shape
    out = torch.empty((M, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.gr_neighbor_pool(_lib.ptr(xx), N, C, _lib.ptr(nb), M, H, mode, _lib.ptr(out), _lib.stream_ptr(dev)))
    return out if out_device.type == "cuda" else out.to(out_device)


@torch.no_grad()
def maxpool(x, neighbor_indices):
    """kpconv/functional.py:54-67."""
    return _pool(x, neighbor_indices, 0)


@torch.no_grad()
def nearest_upsample(x, upsample_indices):
    """kpconv/functional.py:6-22 (only the first neighbour column is used)."""
    return _pool(x, upsample_indices, 1)
