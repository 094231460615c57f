"""fp32-MFMA kernels: achieved TFLOP/s vs the gfx950 dense fp32 matrix peak (157.3 TFLOP/s, MI355X_MICROARCH.md).
Dev tool + the command profiled for profiles/*_mfma.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import ops
from gaussreg_amd.embedding import GeometricStructureEmbedding
from gaussreg_amd.kpconv import KPConv

PEAK = 157.3


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


g = torch.Generator(device="cuda").manual_seed(0)
rows = []
for n, c in ((767, 256), (8192, 256)):
    x = torch.randn(n, c, device="cuda", generator=g)
    y = torch.randn(n, c, device="cuda", generator=g)
    t = timeit(lambda: ops.pairwise_distance(x, y))
    rows.append((f"pairwise_distance {n}x{n}x{c}", 2.0 * n * n * c, t))
# the configs[4] shape: 64 pairs x 767 superpoints in one launch -- SuperPointMatching (distance + exp on the MFMA kernel) and the bare batch
from gaussreg_amd.matching import SuperPointMatching
fb = torch.nn.functional.normalize(torch.randn(64 * 2 * 767, 256, device="cuda", generator=g), dim=1)
spm = SuperPointMatching(256)
t = timeit(lambda: spm.forward_batch(fb, [767] * 128), 5)
rows.append(("superpoint_matching_batch 64 x 767x767x256 (whole op)", 2.0 * 64 * 767 * 767 * 256, t))
xb = torch.randn(64, 767, 256, device="cuda", generator=g)
t = timeit(lambda: ops.pairwise_distance(xb, xb), 5)
rows.append(("pairwise_distance batch 64 x 767x767x256", 2.0 * 64 * 767 * 767 * 256, t))
pc = torch.rand(1, 767, 3, device="cuda", generator=g) * 5
for fp32 in (True, False):
    gse = GeometricStructureEmbedding(256, 0.2, 15, 3, fp32_mfma=fp32).cuda()
    t = timeit(lambda: gse(pc), 5)
    rows.append((f"geo_embedding N=767 C=256 k=3 ({'fp32 MFMA' if fp32 else 'split-bf16 x6 MFMA, fp32-equivalent flops'})", 2.0 * 767 * 767 * 256 * 256 * 4, t))
N, H, C = 28020, 43, 256
spt = torch.rand(N, 3, device="cuda", generator=g)
nb = torch.randint(0, N + 1, (N, H), device="cuda")
f = torch.relu(torch.randn(N, C, device="cuda", generator=g))
conv = KPConv(C, C, 15, 0.25, 0.2, kernel_points=torch.randn(15, 3) * 0.1).cuda()
t = timeit(lambda: conv(f, spt, spt, nb), 5)
rows.append(("kpconv 28020 pts H=43 256->256 (gather + 15x256x256 GEMM)", 2.0 * N * 15 * C * C + 2.0 * N * H * 15 * C, t))
for name, flop, t in rows:
    print(f"{name}: {t*1e3:.3f} ms  {flop/t/1e12:.1f} TFLOP/s  = {100*flop/t/1e12/PEAK:.1f}% of fp32 MFMA peak")
