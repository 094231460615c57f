"""HIP operators vs the SAME computation written with stock torch ops on the same GPU (what the
reference executes: its matching / KPConv / Sinkhorn code is plain PyTorch).  Demo shapes."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd.matching import SuperPointMatching, PointMatching
from gaussreg_amd.ops import point_to_node_partition, pairwise_distance
from gaussreg_amd.sinkhorn import LearnableLogOptimalTransport
from gaussreg_amd.kpconv import KPConv

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
# ---- SuperPointMatching 767 x 767 x 256
ref = torch.nn.functional.normalize(R(767, 256), dim=1); src = torch.nn.functional.normalize(R(767, 256), dim=1)
def spm_torch():
    s = torch.exp(-(2.0 - 2.0 * ref @ src.T).clamp(min=0))
    s = (s / s.sum(1, keepdim=True)) * (s / s.sum(0, keepdim=True))
    sc, idx = s.view(-1).topk(256); return idx // 767, idx % 767, sc
spm = SuperPointMatching(256)
print(f"SuperPointMatching 767x767x256: HIP {timeit(lambda: spm(ref, src)):.3f} ms  torch {timeit(spm_torch):.3f} ms")
# ---- point_to_node 24745 / 767 / 128
pts = torch.rand(24745, 3, device="cuda", generator=g) * 3; nodes = pts[torch.randperm(24745, device="cuda")[:767]]
def p2n_torch():
    d = ((nodes ** 2).sum(1)[:, None] - 2 * nodes @ pts.T + (pts ** 2).sum(1)[None]).clamp(min=0)
    p2n = d.min(0)[1]
    m = torch.zeros_like(d, dtype=torch.bool); m[p2n, torch.arange(24745, device="cuda")] = True
    d.masked_fill_(~m, 1e12); idx = d.topk(128, dim=1, largest=False)[1]; return p2n, idx
print(f"point_to_node_partition 24745/767/128: HIP {timeit(lambda: point_to_node_partition(pts, nodes, 128)):.3f} ms  torch {timeit(p2n_torch):.3f} ms")
# ---- PointMatching 256 x 128 x 128
P, K = 256, 128
score = torch.log_softmax(R(P, K, K) * 3, 2); rm = torch.rand(P, K, device="cuda", generator=g) > 0.3; sm = torch.rand(P, K, device="cuda", generator=g) > 0.3
rp, sp = R(P, K, 3), R(P, K, 3); ri = torch.randint(0, 30000, (P, K), device="cuda"); si = torch.randint(0, 30000, (P, K), device="cuda"); gs = torch.rand(P, device="cuda")
def pm_torch():
    e = torch.exp(score); mask = rm[:, :, None] & sm[:, None, :]
    bi = torch.arange(P, device="cuda")
    v, i = e.topk(3, dim=2); a = torch.zeros_like(e); a[bi.view(P,1,1).expand(-1,K,3), torch.arange(K, device="cuda").view(1,K,1).expand(P,-1,3), i] = v
    v2, i2 = e.topk(3, dim=1); b = torch.zeros_like(e); b[bi.view(P,1,1).expand(-1,3,K), i2, torch.arange(K, device="cuda").view(1,1,K).expand(P,3,-1)] = v2
    c = (a > 0.05) & (b > 0.05) & mask; bb, ii, jj = torch.nonzero(c, as_tuple=True)
    return rp[bb, ii], sp[bb, jj], ri[bb, ii], si[bb, jj], e[bb, ii, jj]
pm = PointMatching(3)
print(f"PointMatching 256x128x128: HIP {timeit(lambda: pm(rp, sp, rm, sm, ri, si, score, gs)):.3f} ms  torch {timeit(pm_torch):.3f} ms")
# ---- Sinkhorn 256 x 128 x 128, 100 iterations
ot = LearnableLogOptimalTransport(100).cuda(); sc = R(P, K, K)
def sk_torch():
    pad = torch.full((P, K + 1, K + 1), 1.0, device="cuda"); pad[:, :K, :K] = sc
    prm = torch.zeros(P, K + 1, dtype=torch.bool, device="cuda"); prm[:, :K] = ~rm
    pcm = torch.zeros(P, K + 1, dtype=torch.bool, device="cuda"); pcm[:, :K] = ~sm
    pad.masked_fill_(prm[:, :, None] | pcm[:, None, :], -1e12)
    nr, nc = rm.float().sum(1), sm.float().sum(1); norm = -torch.log(nr + nc)
    mu = norm[:, None].expand(P, K + 1).clone(); mu[:, K] = torch.log(nc) + norm; mu[prm] = -1e12
    nu = norm[:, None].expand(P, K + 1).clone(); nu[:, K] = torch.log(nr) + norm; nu[pcm] = -1e12
    u, v = torch.zeros_like(mu), torch.zeros_like(nu)
    for _ in range(100):
        u = mu - torch.logsumexp(pad + v[:, None, :], 2); v = nu - torch.logsumexp(pad + u[:, :, None], 1)
    return pad + u[:, :, None] + v[:, None, :] - norm[:, None, None]
print(f"Sinkhorn 256x128x128 (100 it): HIP {timeit(lambda: ot(sc, rm, sm), 10):.3f} ms  torch {timeit(sk_torch, 5):.3f} ms")
# ---- KPConv stage-2 shape: M = N = 28020, H = 43, 256 -> 256
N, H, C = 28020, 43, 256
spt = torch.rand(N, 3, device="cuda", generator=g); nb = torch.randint(0, N + 1, (N, H), device="cuda")
f = torch.relu(R(N, C)); kp = R(15, 3) * 0.1
conv = KPConv(C, C, 15, 0.25, 0.2, kernel_points=kp.cpu()).cuda()
def kp_torch():
    s2 = torch.cat([spt, torch.zeros_like(spt[:1]) + 1e6], 0); nbp = s2[nb] - spt[:, None]
    w = (1 - ((nbp[:, :, None] - kp) ** 2).sum(3).sqrt() / 0.2).clamp(min=0).transpose(1, 2)
    f2 = torch.cat([f, torch.zeros_like(f[:1])], 0); nf = f2[nb]
    o = (torch.matmul(w, nf).permute(1, 0, 2) @ conv.weights).sum(0)
    num = (nf.sum(-1) > 0).sum(-1).clamp(min=1); return o / num[:, None]
print(f"KPConv 28020 pts, H=43, 256->256: HIP {timeit(lambda: conv(f, spt, spt, nb), 10):.3f} ms  torch {timeit(kp_torch, 5):.3f} ms")
# ---- GeometricStructureEmbedding N = 767, C = 256, k = 3 (demo superpoint count)
from gaussreg_amd.embedding import GeometricStructureEmbedding
from gaussreg_amd.rpe_attention import RPEMultiHeadAttention
N, C = 767, 256
gse = GeometricStructureEmbedding(C, 0.2, 15, 3).cuda(); pc = torch.rand(1, N, 3, device="cuda", generator=g) * 5
def gse_torch():  # geotransformer.py:26-73 with stock ops
    p = pc; xy = p @ p.transpose(1, 2); x2 = (p ** 2).sum(-1)
    dist = (x2[:, :, None] - 2 * xy + x2[:, None, :]).clamp(min=0).sqrt(); d_idx = dist / 0.2
    knn = dist.topk(4, dim=2, largest=False)[1][:, :, 1:]
    kp_ = p[0][knn[0]][None]; ref = kp_ - p[:, :, None]; anc = p[:, None] - p[:, :, None]
    ref = ref[:, :, None].expand(1, N, N, 3, 3); anc = anc[:, :, :, None].expand(1, N, N, 3, 3)
    a_idx = torch.atan2(torch.linalg.norm(torch.cross(ref, anc, dim=-1), dim=-1), (ref * anc).sum(-1)) * gse.factor_a
    return gse.proj_d(gse.embedding(d_idx)) + gse.proj_a(gse.embedding(a_idx)).max(dim=3)[0]
with torch.no_grad():
    print(f"GeometricStructureEmbedding N=767 C=256 k=3: HIP {timeit(lambda: gse(pc), 5):.3f} ms  torch {timeit(gse_torch, 3):.3f} ms")
    emb = gse(pc); att = RPEMultiHeadAttention(C, 4).cuda(); x = R(1, N, C)
    def att_torch():  # rpe_transformer.py:51-72 as written
        q = att.proj_q(x).view(1, N, 4, 64).permute(0, 2, 1, 3); k = att.proj_k(x).view(1, N, 4, 64).permute(0, 2, 1, 3)
        v = att.proj_v(x).view(1, N, 4, 64).permute(0, 2, 1, 3); p = att.proj_p(emb).view(1, N, N, 4, 64).permute(0, 3, 1, 2, 4)
        s = (torch.einsum('bhnc,bhnmc->bhnm', q, p) + torch.einsum('bhnc,bhmc->bhnm', q, k)) / 8.0
        s = torch.softmax(s, -1); return torch.matmul(s, v)
    print(f"RPEMultiHeadAttention N=767 C=256 H=4: HIP+torch {timeit(lambda: att(x, x, x, emb), 5):.3f} ms  torch {timeit(att_torch, 3):.3f} ms")
