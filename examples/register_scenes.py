#!/usr/bin/env python3
"""Coarse registration of two Gaussian-Splatting scenes, end to end on one MI355X -- what GaussReg's
experiments/geotransformer.gaussian_splatting.indoor/demo.py (+ the fusion step of gs_fusion.py) does, with every stage on
the GPU through gaussreg_amd:

    GS .ply -> opacity / percentile filter -> FPS to --num_sample -> [opacity, SH colour] features -> volume normalisation
    -> 5-level pyramid (collate) -> GeoTransformer (KPConvFPN, GeometricTransformer, matching, Sinkhorn, LGR, RANSAC)
    -> similarity transform between the ORIGINAL scenes -> estimated_transform.npz, fused scene (gaussian_fuse)
    -> (--render N) N views of the fused scene through diff_gaussian_rasterization, written as .ppm  (the "render + fuse" of
       gs_fusion.py's use: BASELINE.json configs[3])

    python examples/register_scenes.py --ref_file A/point_cloud.ply --src_file B/point_cloud.ply --weights ckpt.pth.tar
    python examples/register_scenes.py --synthetic            # two views of one synthetic scene, random-init network

Without --weights the network runs on random weights (the stages run, the estimate is meaningless); the reference's
checkpoint format (`state_dict["model"]`, demo.py:141-142) loads as is.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

from gaussreg_amd import gs_io, gs_points, synthetic  # noqa: E402
from gaussreg_amd.data import registration_collate_fn_stack_mode  # noqa: E402
from gaussreg_amd.model import create_model, make_cfg  # noqa: E402

NEIGHBOR_LIMITS = [89, 30, 43, 49, 49]  # demo.py:136


def synthetic_scene_pair(out_dir, n=400_000, seed=0):
    """Two GS .ply files holding the same synthetic scene, the second one moved by a known similarity transform."""
    g = synthetic.gaussians_c2(n, seed, sh_degree=3)
    rec = np.zeros((n, 62), np.float32)
    rec[:, 0:3] = g["means3D"] * 2.0                       # a few metres across
    rec[:, 6:9] = g["shs"][:, 0, :]
    rec[:, 9:54] = np.transpose(g["shs"][:, 1:, :], (0, 2, 1)).reshape(n, 45)
    op = np.clip(g["opacities"][:, 0], 1e-4, 1 - 1e-4)
    rec[:, 54] = np.log(op / (1 - op)) + 2.0               # most of them above the 0.7 cut
    rec[:, 55:58] = np.log(g["scales"])
    rec[:, 58:62] = g["rotations"]
    a = 0.35
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = 1.1 * R
    T[:3, 3] = [0.4, -0.3, 0.2]
    ref_path, src_path = os.path.join(out_dir, "scene_ref.ply"), os.path.join(out_dir, "scene_src.ply")
    gs_io.write_gs_ply(ref_path, rec)
    # the same Gaussians expressed in the frame of the second capture: x_src = T^-1 x_ref (positions, sizes, orientations;
    # the SH coefficients are left as they are -- they only colour the input features here)
    inv = np.linalg.inv(T)
    s_inv = np.cbrt(np.linalg.det(inv[:3, :3]))
    Rm = inv[:3, :3] / s_inv
    moved = rec.copy()
    moved[:, 0:3] = rec[:, 0:3] @ inv[:3, :3].T + inv[:3, 3]
    moved[:, 55:58] = rec[:, 55:58] + np.log(s_inv)
    half = 0.5 * np.arctan2(Rm[1, 0], Rm[0, 0])                      # Rm is a rotation about z
    qz = np.array([np.cos(half), 0.0, 0.0, np.sin(half)])            # (w, x, y, z)
    w1, x1, y1, z1 = qz
    w2, x2, y2, z2 = rec[:, 58], rec[:, 59], rec[:, 60], rec[:, 61]
    moved[:, 58] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    moved[:, 59] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    moved[:, 60] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
    moved[:, 61] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    gs_io.write_gs_ply(src_path, moved.astype(np.float32))
    return ref_path, src_path, T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src_file")
    ap.add_argument("--ref_file")
    ap.add_argument("--output_path", default="demo_outputs")
    ap.add_argument("--weights")
    ap.add_argument("--num_sample", type=int, default=30000)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--render", type=int, default=0, help="render this many views of the fused scene (orbit around its centre)")
    ap.add_argument("--render_size", default="640x480")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    os.makedirs(args.output_path, exist_ok=True)
    T_gt = None
    if args.synthetic:
        args.ref_file, args.src_file, T_gt = synthetic_scene_pair(args.output_path)
    elif not (args.ref_file and args.src_file):
        ap.error("--ref_file and --src_file (or --synthetic) are required")
    t0 = time.perf_counter()
    rec_ref = torch.from_numpy(gs_io.read_gs_ply(args.ref_file)).to(dev)
    rec_src = torch.from_numpy(gs_io.read_gs_ply(args.src_file)).to(dev)
    rp, rf, _ = gs_points.extract_points(rec_ref, args.num_sample)
    sp, sf, _ = gs_points.extract_points(rec_src, args.num_sample)
    pair = gs_points.normalize_pair(rp, rf, sp, sf)
    cfg = make_cfg()
    data = registration_collate_fn_stack_mode([pair], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                              cfg.backbone.init_radius, NEIGHBOR_LIMITS, device=dev)
    torch.manual_seed(0)
    model = create_model(cfg).to(dev).eval()
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location=dev)["model"])
    else:
        print("no --weights: random-init network, the estimate below is not meaningful", file=sys.stderr)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = model(data)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    T = gs_points.denormalize_transform(out["estimated_transform"], pair["ref_center"], pair["src_center"],
                                        pair["ref_adjust_scale"], pair["src_adjust_scale"])
    np.savez(os.path.join(args.output_path, "estimated_transform.npz"), estimated_transform=T)
    fused = gs_io.gaussian_fuse_records(rec_ref, rec_src, T)
    gs_io.write_gs_ply(os.path.join(args.output_path, "fused.ply"), fused.cpu().numpy())
    print(f"points: ref {rp.shape[0]}, src {sp.shape[0]}; superpoints {out['ref_points_c'].shape[0]} / {out['src_points_c'].shape[0]}; "
          f"correspondences {out['ref_corr_points'].shape[0]}")
    print(f"load + sample + collate {1e3 * (t1 - t0):.1f} ms, network {1e3 * (t2 - t1):.1f} ms; fused scene: {fused.shape[0]} Gaussians")
    print("estimated transform (src -> ref):\n", np.array_str(T, precision=4, suppress_small=True))
    if T_gt is not None:
        print("ground truth:\n", np.array_str(T_gt, precision=4, suppress_small=True))
    if args.render > 0:
        render_fused(fused, args.render, args.render_size, args.output_path)
    return T


def render_fused(fused, n_views, size, output_path):
    """Render the fused scene from `n_views` cameras on a ring around its centre through the drop-in rasterizer API
    (diff_gaussian_rasterization.GaussianRasterizationSettings / GaussianRasterizer) and write fused_view_XX.ppm."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    W, H = (int(v) for v in size.lower().split("x"))
    parts = gs_io.split_records(fused)
    xyz = parts["means3D"]
    centre = xyz.mean(0)
    radius = float((xyz - centre).norm(dim=1).quantile(0.9)) * 2.2 + 1e-3
    c_np = centre.double().cpu().numpy()
    t0 = time.perf_counter()
    for k in range(n_views):
        ang = 2 * np.pi * k / n_views
        eye = c_np + radius * np.array([np.cos(ang), 0.25, np.sin(ang)])
        z = (c_np - eye) / np.linalg.norm(c_np - eye)          # camera looks along +z (3DGS convention)
        x = np.cross([0.0, 1.0, 0.0], z)
        x = x / np.linalg.norm(x) if np.linalg.norm(x) > 1e-6 else np.array([1.0, 0.0, 0.0])
        cam = synthetic.camera(W, H, 60.0, np.stack([x, np.cross(z, x), z], 1), eye)
        rs = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], torch.zeros(3), 1.0,
                                           torch.from_numpy(cam["viewmatrix"]), torch.from_numpy(cam["projmatrix"]), 3,
                                           torch.from_numpy(cam["campos"]), False, False)
        img, _ = GaussianRasterizer(rs)(xyz, None, parts["opacities"], shs=parts["shs"], scales=parts["scales"],
                                        rotations=parts["rotations"])
        rgb = (img.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
        with open(os.path.join(output_path, f"fused_view_{k:02d}.ppm"), "wb") as fh:
            fh.write(b"P6 %d %d 255\n" % (W, H))
            fh.write(rgb.tobytes())
    torch.cuda.synchronize()
    print(f"rendered {n_views} views of the fused scene at {W}x{H} in {1e3 * (time.perf_counter() - t0):.1f} ms (incl. file output)")


if __name__ == "__main__":
    main()
