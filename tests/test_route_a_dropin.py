"""INTEGRATION.md route A: this repo FIRST on sys.path, GaussReg's checkout behind it.  The import blocks of the
reference's own model.py:1-17 and demo.py:1-19 must then execute unchanged: the hot-path names resolve to gaussreg_amd,
everything this repo does not replace (modules.registration, utils.torch / open3d / registration, engine, ...) resolves to
GaussReg's own files through the chained package paths (gaussreg_amd/_alias.py).

Build-container only: skipped where /root/reference is absent (the GPU box).  Nothing of the reference is copied: the
two import blocks are read from the reference's files at run time, in a child interpreter that writes no bytecode and
whose `ensure_dir` is neutralised (config.py:24-29 creates output directories next to the checkout on import)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference"
EXP = os.path.join(REF, "experiments", "geotransformer.gaussian_splatting.indoor")

CHILD = textwrap.dedent('''
    import json, os, sys, types
    ROOT, REF, EXP = sys.argv[1:4]
    sys.path[:0] = [ROOT, REF, EXP]

    class Any:                                     # stand-in for third-party packages that do no arithmetic on this path
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return Any()
        def __call__(self, *a, **k): return Any()

    def stub(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        def __setattr__(self, k, v): self[k] = v

    stub("IPython", embed=lambda *a, **k: None)
    stub("ipdb", set_trace=lambda *a, **k: None)
    o3d = stub("open3d")
    for sub in ("geometry", "utility", "io", "visualization", "registration", "pipelines"):
        setattr(o3d, sub, Any())
    stub("plyfile", PlyData=Any, PlyElement=Any)
    stub("easydict", EasyDict=EasyDict)
    stub("coloredlogs", install=lambda *a, **k: None)
    stub("cv2")
    import geotransformer.utils.common as common   # GaussReg's own file, through the chained path
    common.ensure_dir = lambda path: None          # config.py would create output/ directories inside the checkout

    ns = {}
    exec("\\n".join(open(os.path.join(EXP, "model.py")).read().split("\\n")[:17]), ns)
    exec("\\n".join(open(os.path.join(EXP, "demo.py")).read().split("\\n")[:19]), ns)
    import geotransformer.modules.ops as ops
    import geotransformer.modules.ops.pointcloud_partition as pp
    import geotransformer.modules.kpconv.modules as km
    from geotransformer.modules.ops import radius_search, grid_subsample
    mod = lambda o: o.__module__
    out = {k: mod(ns[k]) for k in ("point_to_node_partition", "index_select", "get_node_correspondences",
                                   "LearnableLogOptimalTransport", "GeometricTransformer", "SuperPointMatching",
                                   "SuperPointTargetGenerator", "LocalGlobalRegistration", "KPConvFPN",
                                   "registration_with_ransac_from_correspondences", "registration_collate_fn_stack_mode",
                                   "to_cuda", "compute_registration_error_w_scale", "create_model")}
    out["radius_search"], out["grid_subsample"] = mod(radius_search), mod(grid_subsample)
    out["ops_names"] = sorted(n for n in dir(ops) if not n.startswith("_"))
    out["knn_partition"], out["pp.point_to_node_partition"] = mod(pp.knn_partition), mod(pp.point_to_node_partition)
    out["ConvBlock"] = mod(km.ConvBlock)
    out["fpsample"] = ns["fpsample"].__file__
    out["geotransformer_path"] = list(sys.modules["geotransformer"].__path__)
    # GaussReg's own create_model(cfg) (model.py:225-227) over GaussReg's own backbone.py, assembled from this repo's blocks
    m = ns["create_model"](ns["make_cfg"]())
    out["model"] = {k: type(v).__module__ for k, v in (("model", m), ("backbone", m.backbone), ("encoder1_1", m.backbone.encoder1_1),
                    ("transformer", m.transformer), ("coarse_matching", m.coarse_matching), ("fine_matching", m.fine_matching),
                    ("optimal_transport", m.optimal_transport))}
    out["params"] = sum(p.numel() for p in m.parameters())
    print("RESULT " + json.dumps(out))
''')

# geotransformer/modules/ops/__init__.py:1-22 of GaussReg
OPS_NAMES = ["grid_subsample", "index_select", "pairwise_distance", "get_point_to_node_indices", "point_to_node_partition",
             "knn_partition", "ball_query_partition", "radius_search", "apply_transform", "apply_rotation", "inverse_transform",
             "skew_symmetric_matrix", "rodrigues_rotation_matrix", "rodrigues_alignment_matrix",
             "get_transform_from_rotation_translation", "get_rotation_translation_from_transform",
             "get_rotation_translation_from_transform_w_scale", "vector_angle", "rad2deg", "deg2rad"]


@pytest.mark.skipif(not os.path.isdir(EXP), reason="GaussReg checkout not mounted (build container only)")
def test_reference_import_blocks_run_over_the_alias_packages(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, REF, EXP], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    ours = ("point_to_node_partition", "index_select", "LearnableLogOptimalTransport", "GeometricTransformer", "SuperPointMatching",
            "LocalGlobalRegistration", "registration_collate_fn_stack_mode", "radius_search", "grid_subsample",
            "pp.point_to_node_partition", "ConvBlock")
    for k in ours:
        assert out[k].startswith("gaussreg_amd"), (k, out[k])
    assert out["get_node_correspondences"] == "geotransformer.modules.registration.matching"
    assert out["SuperPointTargetGenerator"] == "geotransformer.modules.geotransformer.superpoint_target"
    assert out["to_cuda"] == "geotransformer.utils.torch" and out["compute_registration_error_w_scale"] == "geotransformer.utils.registration"
    assert out["registration_with_ransac_from_correspondences"] == "geotransformer.utils.open3d"
    assert out["KPConvFPN"] == "backbone" and out["create_model"] == "model"      # GaussReg's own experiment files
    assert out["knn_partition"].endswith("pointcloud_partition__upstream")
    assert os.path.samefile(out["fpsample"], os.path.join(ROOT, "fpsample.py"))
    assert set(OPS_NAMES) <= set(out["ops_names"]), sorted(set(OPS_NAMES) - set(out["ops_names"]))
    assert os.path.samefile(out["geotransformer_path"][0], os.path.join(ROOT, "geotransformer"))
    assert os.path.samefile(out["geotransformer_path"][1], os.path.join(REF, "geotransformer"))
    # the reference's model class, built from this repo's blocks, with the reference's parameter count (28.4 M)
    assert out["model"]["model"] == "model" and out["model"]["backbone"] == "backbone"
    for k in ("encoder1_1", "transformer", "coarse_matching", "fine_matching", "optimal_transport"):
        assert out["model"][k].startswith("gaussreg_amd"), (k, out["model"][k])
    assert out["params"] == 28411201


def test_alias_packages_stand_alone_without_the_reference(tmp_path):
    """Without GaussReg's checkout on the path the alias packages still import and carry the hot-path surface."""
    code = ("import sys; sys.path.insert(0, %r); import geotransformer.modules.ops as o, geotransformer.modules.geotransformer as g, "
            "geotransformer.modules.kpconv, geotransformer.modules.transformer, geotransformer.modules.sinkhorn, geotransformer.utils.data, "
            "geotransformer.ext; assert len(sys.modules['geotransformer'].__path__) == 1; "
            "assert all(hasattr(o, n) for n in ('grid_subsample', 'radius_search', 'pairwise_distance', 'point_to_node_partition', 'index_select')); "
            "assert not hasattr(g, 'SuperPointTargetGenerator') and hasattr(g, 'SuperPointMatching')" % ROOT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
