"""Where one forward of gaussreg_amd.model.GeoTransformer spends its time (2 x 30 000-pt pair, random weights)."""
import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
from gaussreg_amd.data import precompute_data_stack_mode
from gaussreg_amd.model import GeoTransformer, make_cfg
dev = torch.device("cuda", 0)
r, s, _ = pair_pipeline.synthetic_room_pair(0, 30000, dev)
pts = torch.cat([r, s]).contiguous()
d = precompute_data_stack_mode(pts, torch.tensor([30000, 30000]), 5, 0.025, 0.0625, [89, 30, 43, 49, 49])
d["features"] = torch.rand(pts.shape[0], 4, device=dev)
torch.manual_seed(0)
net = GeoTransformer(make_cfg()).to(dev).eval()
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"forward            {t(lambda: net(d)):8.2f} ms")
print(f"backbone           {t(lambda: net.backbone(d['features'], d)):8.2f} ms")
fl = net.backbone(d["features"], d)
nc = int(d["lengths"][-1][0]); pc = d["points"][-1]
rc, sc = pc[:nc][None], pc[nc:][None]
fr, fs = fl[-1][:nc][None], fl[-1][nc:][None]
print(f"transformer        {t(lambda: net.transformer(rc, sc, fr, fs)):8.2f} ms")
print(f"  embedding x2     {t(lambda: (net.transformer.embedding(rc), net.transformer.embedding(sc))):8.2f} ms")
e0, e1 = net.transformer.embedding(rc), net.transformer.embedding(sc)
x0, x1 = net.transformer.in_proj(fr), net.transformer.in_proj(fs)
print(f"  6 blocks         {t(lambda: net.transformer.transformer(x0, x1, e0, e1)):8.2f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    net.backbone(d["features"], d); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
