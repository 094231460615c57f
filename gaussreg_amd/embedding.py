"""Mirror of geotransformer/modules/geotransformer/geotransformer.py:9-73 (GeometricStructureEmbedding) and
geotransformer/modules/transformer/positional_embedding.py:8-34 (SinusoidalPositionalEmbedding), inference only.
`forward` runs one fused HIP kernel per cloud (gaussreg_amd/csrc/geo_embedding.hip); state-dict keys are the
reference's (`embedding.div_term` is a buffer, `proj_d.{weight,bias}`, `proj_a.{weight,bias}`)."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib


class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        if d_model % 2 != 0:
            raise ValueError(f'Sinusoidal positional encoding with odd d_model: {d_model}')
        self.d_model = d_model
        div_indices = torch.arange(0, d_model, 2).float()
        div_term = torch.exp(div_indices * (-np.log(10000.0) / d_model))
        self.register_buffer('div_term', div_term)

    @torch.no_grad()
    def forward(self, emb_indices):
        """(*) -> (*, d_model), (sin, cos) interleaved; plain tensor ops (the fused kernel never materialises this)."""
        omegas = emb_indices.reshape(-1, 1, 1) * self.div_term.view(1, -1, 1)
        emb = torch.cat([torch.sin(omegas), torch.cos(omegas)], dim=2)
        return emb.view(*emb_indices.shape, self.d_model)


class GeometricStructureEmbedding(nn.Module):
    TABLE_INV_H = 32.0        # table step 1/32 of an embedding index: interpolation error ~ 2e-8
    TABLE_X_MAX_D = 256.0     # distance indices up to 256 (= 51 m at sigma_d 0.2) come from the table, beyond: direct evaluation

    def __init__(self, hidden_dim, sigma_d, sigma_a, angle_k, reduction_a='max', fp32_mfma=False, mode="table"):
        """mode="table" (default): the two projections are tabulated as functions of their scalar index once per set of
        weights (fp64) and interpolated in the kernel -- gr_geo_embedding_table; "gemm": the fused sinusoid -> matrix-core
        kernel (`fp32_mfma=True` runs it on fp32 MFMAs instead of the split-bf16 scheme)."""
        super().__init__()
        if mode not in ("table", "gemm"):
            raise ValueError("mode must be 'table' or 'gemm'")
        self.mode = "gemm" if fp32_mfma else mode
        self._tables = None
        self.fp32_mfma = bool(fp32_mfma)
        self.sigma_d = sigma_d
        self.sigma_a = sigma_a
        self.factor_a = 180.0 / (self.sigma_a * np.pi)
        self.angle_k = angle_k
        self.embedding = SinusoidalPositionalEmbedding(hidden_dim)
        self.proj_d = nn.Linear(hidden_dim, hidden_dim)
        self.proj_a = nn.Linear(hidden_dim, hidden_dim)
        self.reduction_a = reduction_a
        if self.reduction_a not in ['max', 'mean']:
            raise ValueError(f'Unsupported reduction mode: {self.reduction_a}.')

    @torch.no_grad()
    def _function_tables(self, dev):
        """F_d(x) = proj_d(embedding(x)) and F_a(x) = proj_a(embedding(x)) on the grid x = (j - 1) / TABLE_INV_H, evaluated
        in fp64 and stored as fp32 (rows x hidden_dim); rebuilt when a weight changes (version counters) or moves."""
        ps = (self.proj_d.weight, self.proj_d.bias, self.proj_a.weight, self.proj_a.bias, self.embedding.div_term)
        stamp = _lib.tensor_stamp(ps)          # None: inference tensors, no version counters -> rebuild every call
        stamp = None if stamp is None else stamp + (str(dev),)
        if stamp is None or self._tables is None or self._tables[0] != stamp:
            div = self.embedding.div_term.detach().to(dev, torch.float64)

            def table(lin, x_max):
                rows = int(x_max * self.TABLE_INV_H) + 4
                x = (torch.arange(rows, device=dev, dtype=torch.float64) - 1.0) / self.TABLE_INV_H
                om = x[:, None] * div[None, :]
                emb = torch.stack([torch.sin(om), torch.cos(om)], dim=2).reshape(rows, -1)    # (sin, cos) interleaved
                return (emb @ lin.weight.detach().to(dev, torch.float64).t()
                        + lin.bias.detach().to(dev, torch.float64)).to(torch.float32).contiguous()

            x_max_a = float(np.pi * self.factor_a) + 1.0
            self._tables = (stamp, table(self.proj_d, self.TABLE_X_MAX_D), table(self.proj_a, x_max_a))
        return self._tables[1], self._tables[2]

    @torch.no_grad()
    def forward(self, points):
        """points (B, N, 3) -> embeddings (B, N, N, hidden_dim), geotransformer.py:57-73."""
        dev = _lib.require_gpu()
        L = _lib.lib()
        if points.dim() != 3 or points.shape[-1] != 3:
            raise ValueError("points must be (B, N, 3)")
        out_device = points.device
        p = (points if points.is_cuda else points.to(dev)).to(torch.float32).contiguous()
        dev = p.device
        B, N, _ = p.shape
        C = self.proj_d.weight.shape[0]
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        wd, bd, wa, ba, div = f(self.proj_d.weight), f(self.proj_d.bias), f(self.proj_a.weight), f(self.proj_a.bias), \
            f(self.embedding.div_term)
        out = torch.empty((B, N, N, C), dtype=torch.float32, device=dev)
        if self.mode == "table" and C % 4 == 0:
            td, ta = self._function_tables(dev)
            with torch.cuda.device(dev):
                ws = _lib.workspace(dev, L.gr_geo_embedding_workspace_bytes(N, int(self.angle_k)))
                for b in range(B):
                    _lib.check(L.gr_geo_embedding_table(_lib.ptr(p[b]), N, _lib.ptr(td), td.shape[0], _lib.ptr(ta), ta.shape[0],
                                                        float(self.TABLE_INV_H), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(wa),
                                                        _lib.ptr(ba), _lib.ptr(div), C, float(self.sigma_d), float(self.factor_a),
                                                        int(self.angle_k), 1 if self.reduction_a == 'mean' else 0,
                                                        _lib.ptr(out[b]), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
            return out if out_device.type == "cuda" else out.to(out_device)
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_geo_embedding_workspace_bytes(N, int(self.angle_k)))
            for b in range(B):
                _lib.check(L.gr_geo_embedding(_lib.ptr(p[b]), N, _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(wa), _lib.ptr(ba),
                                              _lib.ptr(div), C, float(self.sigma_d), float(self.factor_a),
                                              int(self.angle_k), (1 if self.reduction_a == 'mean' else 0) | (2 if self.fp32_mfma else 0),
                                              _lib.ptr(out[b]), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return out if out_device.type == "cuda" else out.to(out_device)
