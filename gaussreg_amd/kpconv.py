"""Mirror of geotransformer/modules/kpconv (forward, inference): KPConv + maxpool + nearest_upsample on HIP.

`KPConv` keeps the reference's parameters / buffers (`weights` (K,Cin,Cout), optional `bias`, buffer
`kernel_points` (K,3)) so reference checkpoints load with the same state-dict keys (kpconv.py:54-65).
Kernel points: like the reference, a fresh module gets the disposition `k_015_center_3D` (the only one GaussReg uses:
kernel_size 15, config.py:81) scaled by the radius, jittered and rotated about z (kernel_points.py:389-455); the
reference reads those 15 points from a PLY asset through open3d, here they are a table.  Other kernel sizes would need
the reference's kernel optimiser, which is not reproduced: such a module must get `kernel_points` passed in or loaded
from a checkpoint and refuses to run until then.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib

# the 15 points of geotransformer/modules/kpconv/dispositions/k_015_center_3D.ply (unit-radius disposition, first = centre)
K015_CENTER_3D = np.array([
    [0.0, 0.0, 0.0], [-0.49820612, 0.41826797, 0.11736718], [-0.24123565, -0.34214048, -0.5115481],
    [-0.2828808, -0.58614266, 0.11553228], [0.29054036, -0.10093209, -0.585091], [0.42820039, 0.39929883, -0.30681813],
    [-0.63586493, -0.08196441, -0.16090403], [-0.43181082, -0.14729417, 0.47830957], [-0.044666, 0.27973214, 0.59723308],
    [0.22552417, -0.34462544, 0.50794659], [0.63889212, -0.16914906, -0.01190108], [-0.22552415, 0.34462545, -0.50794659],
    [0.49054666, 0.26880703, 0.35219206], [0.25233084, -0.59706653, -0.12951142], [0.03415394, 0.65858341, 0.04513958]])


def load_kernels(radius, num_kpoints, dimension=3, fixed='center'):
    """kernel_points.py:389-455 for the stored disposition: unit kernel + N(0, 0.01) noise, scaled by `radius`, rotated
    by a random angle about z (numpy's global RNG, like the reference).  Returns (K, 3) float32."""
    if (num_kpoints, dimension, fixed) != (15, 3, 'center'):
        raise NotImplementedError("only the k_015_center_3D disposition is available (the kernel-point optimiser of "
                                  "kernel_points.py is not reproduced); pass kernel_points or load a checkpoint")
    theta = np.random.rand() * 2 * np.pi
    c, s_ = np.cos(theta), np.sin(theta)
    R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=np.float32)
    pts = K015_CENTER_3D.astype(np.float32) + np.random.normal(scale=0.01, size=K015_CENTER_3D.shape)
    return np.matmul(radius * pts, R).astype(np.float32)


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
        self._kernel_points_ready = True
        if kernel_points is not None:
            kp = torch.as_tensor(kernel_points).float()
        else:
            try:
                kp = torch.from_numpy(load_kernels(radius, kernel_size, dimension=dimension, fixed='center')).float()
            except NotImplementedError:
                kp = torch.zeros(kernel_size, dimension)  # placeholder until a checkpoint fills it
                self._kernel_points_ready = False
        self.register_buffer('kernel_points', kp)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if prefix + 'kernel_points' in state_dict:
            self._kernel_points_ready = True
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @torch.no_grad()
    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        if not self._kernel_points_ready:
            raise RuntimeError("KPConv.kernel_points is uninitialised: no stored disposition for kernel_size=%d; pass "
                               "kernel_points= or load a state dict that carries them" % self.kernel_size)
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
