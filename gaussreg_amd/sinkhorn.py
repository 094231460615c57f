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
    def forward(self, scores, row_masks=None, col_masks=None, drop_dustbin=False, out=None):
        """scores (B, M, N) -> matching scores (B, M+1, N+1), learnable_sinkhorn.py:20-66.
        Extensions (the reference has neither argument): `drop_dustbin=True` returns (B, M, N), the matrix without its dustbin
        row and column -- what model.py:197-198 slices off right after the call; `out`: a preallocated contiguous float32
        CUDA tensor of that shape to write into."""
        dev = _lib.require_gpu()
        L = _lib.lib()
        out_device = scores.device
        s = (scores if scores.is_cuda else scores.to(dev)).to(torch.float32).contiguous()
        dev = s.device
        B, M, N = s.shape
        rm = None if row_masks is None else row_masks.to(device=dev, dtype=torch.bool).contiguous()
        cm = None if col_masks is None else col_masks.to(device=dev, dtype=torch.bool).contiguous()
        alpha = self.alpha.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        shape = (B, M, N) if drop_dustbin else (B, M + 1, N + 1)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=dev)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous float32 tensor of shape %s on %s" % (shape, dev))
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_sinkhorn_workspace_bytes(B))
            _lib.check(L.gr_sinkhorn(_lib.ptr(s), B, M, N, _lib.ptr(rm), _lib.ptr(cm), _lib.ptr(alpha),
                                     int(self.num_iterations), float(self.inf), int(bool(drop_dustbin)), _lib.ptr(out),
                                     _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr(dev)))
        return out if out_device.type == "cuda" else out.to(out_device)

    def __repr__(self):
        return self.__class__.__name__ + '(num_iterations={})'.format(self.num_iterations)
