"""Mirror of geotransformer/modules/sinkhorn/learnable_sinkhorn.py (forward, inference) on one fused
HIP kernel.  Keeps the learnable `alpha` parameter (state-dict key `alpha`) like the reference."""
import torch
import torch.nn as nn

from . import _lib


class LearnableLogOptimalTransport(nn.Module):
    def __init__(self, num_iterations, inf=1e12):
        super().__init__()
        self.num_iterations = num_iterations
        self.register_parameter('alpha', torch.nn.Parameter(torch.tensor(1.0)))
        self.inf = inf

    @torch.no_grad()
    def forward(self, scores, row_masks=None, col_masks=None):
        """scores (B, M, N) -> matching scores (B, M+1, N+1), learnable_sinkhorn.py:20-66."""
        dev = _lib.require_gpu()
        L = _lib.lib()
        out_device = scores.device
        s = (scores if scores.is_cuda else scores.to(dev)).to(torch.float32).contiguous()
        dev = s.device
        B, M, N = s.shape
        rm = None if row_masks is None else row_masks.to(device=dev, dtype=torch.bool).contiguous()
        cm = None if col_masks is None else col_masks.to(device=dev, dtype=torch.bool).contiguous()
        alpha = self.alpha.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        out = torch.empty((B, M + 1, N + 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_sinkhorn_workspace_bytes(B))
            _lib.check(L.gr_sinkhorn(_lib.ptr(s), B, M, N, _lib.ptr(rm), _lib.ptr(cm), _lib.ptr(alpha),
                                     int(self.num_iterations), float(self.inf), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr(dev)))
        return out if out_device.type == "cuda" else out.to(out_device)

    def __repr__(self):
        return self.__class__.__name__ + '(num_iterations={})'.format(self.num_iterations)
