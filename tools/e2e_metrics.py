"""Margins of the end-to-end model test (tests/test_gpu_model_e2e.py), for DESIGN.md."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import demo_inputs
from gen_golden_ext import room_pair
from helpers import load_golden
from gaussreg_amd.kpconv import KPConv
from gaussreg_amd.model import GeoTransformer, make_cfg
from geotransformer.utils.data import precompute_data_stack_mode
g = load_golden("model_e2e.npz")
ref, src = room_pair(6000, 11)
points = np.concatenate([ref, src]).astype(np.float32)
d = precompute_data_stack_mode(torch.from_numpy(points).cuda(), torch.tensor([6000, 6000]), 5, 0.025, 0.0625, [38, 36, 36, 38, 38])
d["features"] = demo_inputs.backbone_feats(points.shape[0]).cuda()
torch.manual_seed(int(g["seed"]))
net = GeoTransformer(make_cfg())
for m in net.modules():
    if isinstance(m, KPConv):
        m.kernel_points.copy_(torch.from_numpy((demo_inputs.K015 * m.radius).astype(np.float32)))
net = net.cuda().eval()
o = net(d)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): o = net(d)
torch.cuda.synchronize()
print(f"forward: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
fc = o["ref_feats_c"].cpu().numpy()[::3].astype(np.float64)
print("coarse feats |hip-f64|", np.abs(fc - g["ref_feats_c64"]).max(), "|ref32-f64|", np.abs(g["ref_feats_c32"] - g["ref_feats_c64"]).max(), "scale", np.abs(g["ref_feats_c64"]).max())
ff = o["ref_feats_f"].cpu().numpy()[g["feats_f_rows"]].astype(np.float64)
print("fine feats |hip-f64|", np.abs(ff - g["ref_feats_f64"]).max(), "|ref32-f64|", np.abs(g["ref_feats_f32"] - g["ref_feats_f64"]).max(), "scale", np.abs(g["ref_feats_f64"]).max())
mine = set(zip(o["ref_node_corr_indices"].tolist(), o["src_node_corr_indices"].tolist()))
want = set(zip(g["ref_ci32"].tolist(), g["src_ci32"].tolist()))
print("superpoint correspondences in common:", len(mine & want), "of 256")
rows = lambda a, b: set(map(tuple, np.round(np.concatenate([a, b], 1) * 1e5).astype(np.int64).tolist()))
m2, w2 = rows(o["ref_corr_points"].cpu().numpy(), o["src_corr_points"].cpu().numpy()), rows(g["ref_corr32"], g["src_corr32"])
print("point correspondences:", len(m2), "vs", len(w2), "common", len(m2 & w2))
print("LGR transform max diff", np.abs(o["lgr_transform"].cpu().numpy() - g["lgr_transform32"]).max())
print("RANSAC vs LGR max diff", np.abs(o["estimated_transform"].cpu().numpy() - o["lgr_transform"].cpu().numpy()).max())
