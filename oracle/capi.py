"""ctypes front-end to oracle/liboracle.so and oracle/_ref/libgaussreg_ref.so.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  NumPy in, NumPy out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is mounted)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("radius_neighbors_oracle.c", "grid_subsample_oracle.cpp", "rasterizer_oracle.c", "fps_oracle.c", "Makefile")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    need_ref = os.path.isdir("/root/reference/geotransformer/extensions") and (
        force or not os.path.exists(os.path.join(_HERE, "_ref", "libgaussreg_ref.so")))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if need_ref:
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.oracle_radius_neighbors.restype = ctypes.c_int64
        L.oracle_radius_neighbors.argtypes = [c_f32p, ctypes.c_int64, c_f32p, ctypes.c_int64, c_i64p, c_i64p,
                                              ctypes.c_int64, ctypes.c_float, ctypes.c_int,
                                              ctypes.POINTER(c_i64p)]
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_grid_subsampling.restype = ctypes.c_int64
        L.oracle_grid_subsampling.argtypes = [c_f32p, ctypes.c_int64, c_i64p, ctypes.c_int64, ctypes.c_float,
                                              c_f32p, c_i64p]
        L.oracle_exp_det.restype = ctypes.c_float
        L.oracle_exp_det.argtypes = [ctypes.c_float]
        L.oracle_rasterize_forward.restype = ctypes.c_int64
        L.oracle_rasterize_forward.argtypes = (
            [ctypes.c_int] * 3 + [c_f32p, ctypes.c_int, ctypes.c_int] + [c_f32p] * 5 + [ctypes.c_float] +
            [c_f32p] * 5 + [ctypes.c_float, ctypes.c_float, c_f32p, c_i32p])
        L.oracle_raster_preprocess.restype = None
        L.oracle_raster_preprocess.argtypes = (
            [ctypes.c_int] * 3 + [c_f32p] * 5 + [ctypes.c_float] + [c_f32p] * 5 +
            [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float] +
            [c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p])
        L.oracle_mark_visible.restype = None
        L.oracle_mark_visible.argtypes = [ctypes.c_int, c_f32p, c_f32p, ctypes.POINTER(ctypes.c_uint8)]
        L.oracle_fps.restype = ctypes.c_int64
        L.oracle_fps.argtypes = [c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_i64p]
        _LIB = L
    return _LIB


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libgaussreg_ref.so"))


def ref():
    """The reference's own compiled core (oracle/_ref); raises if it was never built."""
    global _REF
    if _REF is None:
        if not have_ref():
            build()
        R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libgaussreg_ref.so"))
        R.ref_radius_neighbors.restype = ctypes.c_int64
        R.ref_radius_neighbors.argtypes = [c_f32p, ctypes.c_int64, c_f32p, ctypes.c_int64, c_i64p, c_i64p,
                                           ctypes.c_int64, ctypes.c_float]
        R.ref_radius_neighbors_fetch.argtypes = [c_i64p, ctypes.c_int64]
        R.ref_grid_subsampling.restype = ctypes.c_int64
        R.ref_grid_subsampling.argtypes = [c_f32p, ctypes.c_int64, c_i64p, ctypes.c_int64, ctypes.c_float,
                                           c_f32p, c_i64p]
        _REF = R
    return _REF


# ---------------------------------------------------------------- radius / grid
def radius_neighbors(q, s, q_lengths, s_lengths, radius, brute=False):
    q, s, ql, sl = _f32(q).reshape(-1, 3), _f32(s).reshape(-1, 3), _i64(q_lengths), _i64(s_lengths)
    out = c_i64p()
    w = lib().oracle_radius_neighbors(_p(q, c_f32p), q.shape[0], _p(s, c_f32p), s.shape[0], _p(ql, c_i64p),
                                      _p(sl, c_i64p), ql.shape[0], float(radius), 1 if brute else 0,
                                      ctypes.byref(out))
    n = q.shape[0] * w
    res = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy().reshape(q.shape[0], w)
    lib().oracle_free(out)
    return res


def grid_subsampling(points, lengths, voxel):
    p, l = _f32(points).reshape(-1, 3), _i64(lengths)
    op = np.empty((max(p.shape[0], 1), 3), np.float32)
    ol = np.empty((l.shape[0],), np.int64)
    m = lib().oracle_grid_subsampling(_p(p, c_f32p), p.shape[0], _p(l, c_i64p), l.shape[0], float(voxel),
                                      _p(op, c_f32p), _p(ol, c_i64p))
    return op[:m].copy(), ol


def ref_radius_neighbors(q, s, q_lengths, s_lengths, radius):
    q, s, ql, sl = _f32(q).reshape(-1, 3), _f32(s).reshape(-1, 3), _i64(q_lengths), _i64(s_lengths)
    w = ref().ref_radius_neighbors(_p(q, c_f32p), q.shape[0], _p(s, c_f32p), s.shape[0], _p(ql, c_i64p),
                                   _p(sl, c_i64p), ql.shape[0], float(radius))
    out = np.empty((q.shape[0], w), np.int64)
    if out.size:
        ref().ref_radius_neighbors_fetch(_p(out, c_i64p), out.size)
    return out


def ref_grid_subsampling(points, lengths, voxel):
    p, l = _f32(points).reshape(-1, 3), _i64(lengths)
    op = np.empty((max(p.shape[0], 1), 3), np.float32)
    ol = np.empty((l.shape[0],), np.int64)
    m = ref().ref_grid_subsampling(_p(p, c_f32p), p.shape[0], _p(l, c_i64p), l.shape[0], float(voxel),
                                   _p(op, c_f32p), _p(ol, c_i64p))
    return op[:m].copy(), ol


# ---------------------------------------------------------------- rasterizer
def exp_det(x):
    return np.array([lib().oracle_exp_det(float(v)) for v in np.atleast_1d(x)], np.float32)


def _opt(a):
    return None if a is None else _f32(a)


def rasterize_forward(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
                      cov3D_precomp=None, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
                      sh_degree=0, scale_modifier=1.0):
    """Returns (color (3,H,W) f32, radii (P,) i32, num_rendered)."""
    m = _f32(means3D).reshape(-1, 3)
    P = m.shape[0]
    shs_, cp, sc, rot, cov = _opt(shs), _opt(colors_precomp), _opt(scales), _opt(rotations), _opt(cov3D_precomp)
    M = 0 if shs_ is None else shs_.reshape(P, -1, 3).shape[1]
    op = _f32(opacities).reshape(-1)
    vm, pm, cam, bgc = _f32(viewmatrix).reshape(-1), _f32(projmatrix).reshape(-1), _f32(campos), _f32(bg)
    color = np.zeros((3, H, W), np.float32)
    radii = np.zeros((max(P, 1),), np.int32)
    R = lib().oracle_rasterize_forward(P, int(sh_degree), M, _p(bgc, c_f32p), W, H, _p(m, c_f32p),
                                       _p(shs_, c_f32p), _p(cp, c_f32p), _p(op, c_f32p), _p(sc, c_f32p),
                                       float(scale_modifier), _p(rot, c_f32p), _p(cov, c_f32p),
                                       _p(vm, c_f32p), _p(pm, c_f32p), _p(cam, c_f32p), float(tanfovx),
                                       float(tanfovy), _p(color, c_f32p), _p(radii, c_i32p))
    if R < 0:
        raise MemoryError("oracle_rasterize_forward")
    return color, radii[:P], int(R)


def raster_preprocess(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
                      cov3D_precomp=None, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
                      sh_degree=0, scale_modifier=1.0):
    m = _f32(means3D).reshape(-1, 3)
    P = m.shape[0]
    shs_, cp, sc, rot, cov = _opt(shs), _opt(colors_precomp), _opt(scales), _opt(rotations), _opt(cov3D_precomp)
    M = 0 if shs_ is None else shs_.reshape(P, -1, 3).shape[1]
    op = _f32(opacities).reshape(-1)
    vm, pm, cam = _f32(viewmatrix).reshape(-1), _f32(projmatrix).reshape(-1), _f32(campos)
    radii = np.zeros((P,), np.int32)
    xy = np.zeros((P, 2), np.float32)
    depths = np.zeros((P,), np.float32)
    co = np.zeros((P, 4), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    tt = np.zeros((P,), np.int32)
    lib().oracle_raster_preprocess(P, int(sh_degree), M, _p(m, c_f32p), _p(shs_, c_f32p), _p(cp, c_f32p),
                                   _p(op, c_f32p), _p(sc, c_f32p), float(scale_modifier), _p(rot, c_f32p),
                                   _p(cov, c_f32p), _p(vm, c_f32p), _p(pm, c_f32p), _p(cam, c_f32p), W, H,
                                   float(tanfovx), float(tanfovy), _p(radii, c_i32p), _p(xy, c_f32p),
                                   _p(depths, c_f32p), _p(co, c_f32p), _p(rgb, c_f32p), _p(tt, c_i32p))
    return dict(radii=radii, xy=xy, depths=depths, conic_opacity=co, rgb=rgb, tiles_touched=tt)


def mark_visible(means3D, viewmatrix):
    m = _f32(means3D).reshape(-1, 3)
    out = np.zeros((m.shape[0],), np.uint8)
    lib().oracle_mark_visible(m.shape[0], _p(m, c_f32p), _p(_f32(viewmatrix).reshape(-1), c_f32p),
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out.astype(bool)


def farthest_point_sampling(points, k, start=0):
    """Sequential exact FPS (oracle/fps_oracle.c); == matching_np.farthest_point_sampling, in C."""
    p = _f32(points)
    out = np.empty(int(k), np.int64)
    got = lib().oracle_fps(_p(p, c_f32p), p.shape[0], int(k), int(start), _p(out, c_i64p))
    assert got == k
    return out
