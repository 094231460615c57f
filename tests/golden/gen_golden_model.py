#!/usr/bin/env python3
"""End-to-end golden for the whole coarse-registration network: the REFERENCE's `GeoTransformer`
(experiments/geotransformer.gaussian_splatting.indoor/model.py:19-248, inference branch) run on the CPU in this container
at GaussReg's configuration (config.py:78-125: KPConvFPN 4 -> 64 -> 256, GeometricTransformer 2048 -> 256, 256 superpoint
correspondences, 128-point patches, 100 Sinkhorn iterations, LocalGlobalRegistration) on the 5-level pyramid of a synthetic
2 x 6 000-point room pair built by the reference C++ core.

Weights are not stored (33 M parameters): both sides build the model under torch.manual_seed(SEED); the file carries an
fp64 checksum of every state-dict tensor.  The Open3D RANSAC at the end of the reference forward (model.py:209-215) cannot
run here (package absent, unseeded sampler) and is replaced by a no-op while generating: everything up to and including
LocalGlobalRegistration is recorded -- twice where it is floating point: the reference's fp32 result and the same model
evaluated in fp64."""
import importlib.util
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import capi  # noqa: E402
from gen_golden_ext import room_pair  # noqa: E402
from gen_golden_demo import import_reference  # noqa: E402
from gen_golden_backbone import pyramid  # noqa: E402
import demo_inputs  # noqa: E402

SEED = 777
EXP = "/root/reference/experiments/geotransformer.gaussian_splatting.indoor"


def ref_cfg():
    """config.py:78-125, the sections model.py reads (make_cfg() itself creates output directories: not imported)."""
    c = SimpleNamespace()
    c.backbone = SimpleNamespace(input_dim=4, output_dim=256, init_dim=64, kernel_size=15, init_radius=2.5 * 0.025,
                                 init_sigma=2.0 * 0.025, group_norm=32)
    c.model = SimpleNamespace(ground_truth_matching_radius=0.05, num_points_in_patch=128, num_sinkhorn_iterations=100)
    c.coarse_matching = SimpleNamespace(num_targets=128, overlap_threshold=0.1, num_correspondences=256, dual_normalization=True)
    c.geotransformer = SimpleNamespace(input_dim=2048, hidden_dim=256, output_dim=256, num_heads=4,
                                       blocks=['self', 'cross', 'self', 'cross', 'self', 'cross'], sigma_d=0.2, sigma_a=15,
                                       angle_k=3, reduction_a='max')
    c.fine_matching = SimpleNamespace(topk=3, acceptance_radius=0.1, mutual=True, confidence_threshold=0.05, use_dustbin=False,
                                      use_global_score=False, correspondence_threshold=3, correspondence_limit=None,
                                      num_refinement_steps=5)
    return c


def main():
    assert os.path.isdir("/root/reference")
    capi.build()
    torch, _ = import_reference()
    torch.set_num_threads(8)
    sys.path.insert(0, EXP)
    import geotransformer.modules.kpconv.kpconv as kp_mod
    kp_mod.load_kernels = lambda radius, k, dimension=3, fixed='center': (demo_inputs.K015 * radius).astype(np.float32)
    spec = importlib.util.spec_from_file_location("ref_model", EXP + "/model.py")
    ref_model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_model)
    ref_model.registration_with_ransac_from_correspondences = lambda *a, **k: np.eye(4)

    # default: 2 x 6 000 points -> model_e2e.npz; `gen_golden_model.py 30000` = the demo size (767-ish superpoints) ->
    # model_e2e_30000.npz
    n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    ref, src = room_pair(n_per, 11)
    points = np.concatenate([ref, src]).astype(np.float32)
    lengths = np.array([n_per, n_per], np.int64)
    pts, lens, nb, sub, up = pyramid(points, lengths)
    feats = demo_inputs.backbone_feats(points.shape[0])

    def data(cast):
        return {"features": cast(feats), "points": [cast(torch.from_numpy(p)) for p in pts],
                "lengths": [torch.from_numpy(l) for l in lens],
                "neighbors": [torch.from_numpy(a.astype(np.int64)) for a in nb],
                "subsampling": [torch.from_numpy(a.astype(np.int64)) for a in sub],
                "upsampling": [torch.from_numpy(a.astype(np.int64)) for a in up]}

    torch.manual_seed(SEED)
    net = ref_model.GeoTransformer(ref_cfg()).eval()
    out = {"seed": np.int64(SEED), "points_sum": np.float64(points.astype(np.float64).sum()),
           "feats_sum": np.float64(feats.double().sum().item()), "level_sizes": np.array([p.shape[0] for p in pts]),
           "param_sums": np.array([float(p.detach().double().sum()) for _, p in sorted(net.state_dict().items())]),
           "param_keys": np.array(sorted(net.state_dict().keys())),
           "param_count": np.int64(sum(p.numel() for p in net.parameters()))}

    def run(model, cast, tag, with_lgr=True):
        captured = {}
        cm, fm = model.coarse_matching.forward, model.fine_matching.forward

        def cm_wrap(*a, **k):
            r = cm(*a, **k)
            captured["node_scores"] = r[2]
            return r

        def fm_wrap(*a, **k):
            if not with_lgr:  # the reference's LocalGlobalRegistration allocates fp32 buffers: it has no fp64 evaluation
                z = torch.zeros((0, 3))
                return z, z, torch.zeros((0,)), torch.eye(4)
            r = fm(*a, **k)
            captured["lgr_transform"] = r[3]
            return r

        model.coarse_matching.forward, model.fine_matching.forward = cm_wrap, fm_wrap
        with torch.no_grad():
            o = model(data(cast))
        model.coarse_matching.forward, model.fine_matching.forward = cm, fm
        res = {f"ref_feats_c{tag}": o["ref_feats_c"].numpy()[::3], f"src_feats_c{tag}": o["src_feats_c"].numpy()[::3],
               f"feats_c_colsum{tag}": np.concatenate([o["ref_feats_c"].double().sum(0).numpy(), o["src_feats_c"].double().sum(0).numpy()]),
               f"ref_ci{tag}": o["ref_node_corr_indices"].numpy(), f"src_ci{tag}": o["src_node_corr_indices"].numpy(),
               f"node_scores{tag}": captured["node_scores"].numpy(),
               f"ms_first{tag}": o["matching_scores"][:4].numpy().astype(np.float32),
               f"ms_rowsum{tag}": torch.logsumexp(o["matching_scores"], dim=2).numpy()}
        if with_lgr:
            res.update({f"ref_corr{tag}": o["ref_corr_points"].numpy(), f"src_corr{tag}": o["src_corr_points"].numpy(),
                        f"corr_scores{tag}": o["corr_scores"].numpy(), f"lgr_transform{tag}": captured["lgr_transform"].numpy()})
        rows = np.arange(0, o["ref_feats_f"].shape[0], 37)
        res[f"feats_f_rows"] = rows
        res[f"ref_feats_f{tag}"] = o["ref_feats_f"].numpy()[rows]
        return res

    out.update(run(net, lambda t: t, "32"))
    out.update(run(net.double(), lambda t: t.double() if t.is_floating_point() else t, "64", with_lgr=False))
    for k in ("ref_feats_c", "node_scores", "ms_first", "ref_feats_f"):
        a, b = out[k + "32"].astype(np.float64), out[k + "64"].astype(np.float64)
        print(k, a.shape, "max |f32 - f64| =", np.abs(a - b).max(), "scale", np.abs(b).max())
    print("node correspondences equal (f32 vs f64):", np.array_equal(out["ref_ci32"], out["ref_ci64"]) and
          np.array_equal(out["src_ci32"], out["src_ci64"]), "corr points:", out["ref_corr32"].shape)
    print("lgr transform\n", out["lgr_transform32"])
    out["n_per_cloud"] = np.int64(n_per)
    path = os.path.join(HERE, "model_e2e.npz" if n_per == 6000 else f"model_e2e_{n_per}.npz")
    np.savez_compressed(path, **out)
    print(os.path.basename(path), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
