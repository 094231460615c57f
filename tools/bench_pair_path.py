"""Per-pair cost of everything this repo puts on the GPU for coarse registration (config[2] stand-in, synthetic
200k-point pair; the learned layers between these operators are the reference's stock PyTorch and are not run here):
FPS 200k -> 30k (x2), data pyramid, geometric embedding (x2), RPE attention score path (3 self blocks x 2 clouds),
SuperPointMatching, point_to_node (x2), Sinkhorn + LocalGlobalRegistration on 256 patches, RANSAC with scale."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
from gen_golden_ext import room_pair
from gaussreg_amd.registration import farthest_point_sampling, registration_with_ransac_from_correspondences
from gaussreg_amd.data import precompute_data_stack_mode
from gaussreg_amd.embedding import GeometricStructureEmbedding
from gaussreg_amd.rpe_attention import RPEMultiHeadAttention
from gaussreg_amd.matching import SuperPointMatching, LocalGlobalRegistration
from gaussreg_amd.ops import point_to_node_partition
from gaussreg_amd.sinkhorn import LearnableLogOptimalTransport


def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
ref, src = room_pair(200000, 0)
big = torch.from_numpy(np.concatenate([ref, src])).cuda()
rows = []
sel = {}
def fps():
    sel["idx"] = farthest_point_sampling(big, [200000, 200000], [30000, 30000])
rows.append(("FPS 2 x (200k -> 30k)", timeit(fps, 3, 1)))
pts = torch.cat([big[:200000][sel["idx"][0]], big[200000:][sel["idx"][1]]]).contiguous()
lens = torch.tensor([30000, 30000])
limits = [89, 30, 43, 49, 49]
out = {}
def pyr():
    out["d"] = precompute_data_stack_mode(pts, lens, 5, 0.025, 0.0625, limits)
rows.append(("data pyramid (4 grid_subsample + 13 radius_search, reference row order)", timeit(pyr)))
pc = out["d"]["points"][-1]; nc = out["d"]["lengths"][-1].tolist()
ref_c, src_c = pc[:nc[0]], pc[nc[0]:]
pf = out["d"]["points"][1]; nf = out["d"]["lengths"][1].tolist()
gse = GeometricStructureEmbedding(256, 0.2, 15, 3).cuda()
emb = {}
def embed():
    emb["r"] = gse(ref_c[None]); emb["s"] = gse(src_c[None])
rows.append((f"GeometricStructureEmbedding x2 (N = {nc[0]}, {nc[1]})", timeit(embed)))
att = RPEMultiHeadAttention(256, 4).cuda()
xr, xs = R(1, nc[0], 256), R(1, nc[1], 256)
def attn():
    for _ in range(3):
        att(xr, xr, xr, emb["r"]); att(xs, xs, xs, emb["s"])
rows.append(("RPEMultiHeadAttention, 3 self blocks x 2 clouds", timeit(attn)))
fr = torch.nn.functional.normalize(R(nc[0], 256), dim=1); fs = torch.nn.functional.normalize(R(nc[1], 256), dim=1)
spm = SuperPointMatching(256)
rows.append(("SuperPointMatching (256 correspondences)", timeit(lambda: spm(fr, fs))))
def p2n():
    a = point_to_node_partition(pf[:nf[0]], ref_c, 128); b = point_to_node_partition(pf[nf[0]:], src_c, 128); return a, b
rows.append((f"point_to_node_partition x2 ({nf[0]} / {nc[0]} / 128)", timeit(p2n)))
P, K = 256, 128
ot = LearnableLogOptimalTransport(100).cuda(); sc = R(P, K, K)
rm = torch.rand(P, K, device="cuda", generator=g) > 0.2; sm = torch.rand(P, K, device="cuda", generator=g) > 0.2
score = {}
def sink():
    score["m"] = ot(sc, rm, sm)
rows.append(("log-Sinkhorn 256 x 128 x 128, 100 iterations", timeit(sink)))
rp = R(P, K, 3); T = torch.eye(4, device="cuda"); sp = rp + 0.01 * R(P, K, 3)
lgr = LocalGlobalRegistration(3, 0.1, True, 0.05, False, False, 3, None, 5)
rows.append(("LocalGlobalRegistration (256 patches, 5 refinement steps)", timeit(lambda: lgr(rp, sp, rm, sm, score["m"][:, :-1, :-1], torch.rand(P, device="cuda")))))
a = torch.rand(5000, 3, device="cuda", generator=g) * 4; bpts = 1.3 * a + 0.01 * R(5000, 3)
rows.append(("RANSAC with scale, 10 000 hypotheses on 5 000 correspondences", timeit(lambda: registration_with_ransac_from_correspondences(a, bpts, None, 0.05, 3, 10000))))
tot = 0.0
for name, ms in rows:
    tot += ms
    print(f"{ms:9.3f} ms  {name}")
print(f"{tot:9.3f} ms  total per pair (operators of this repo only)")
