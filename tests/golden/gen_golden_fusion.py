#!/usr/bin/env python3
"""Golden vector for the GS fusion row: runs the REFERENCE's gs_fusion.gaussian_fuse end to end on two small
synthetic GS .ply files.  `plyfile` is not installed, so a minimal stand-in that does pure I/O (read / write
of the 62-float vertex records; no arithmetic) is registered under that name before importing gs_fusion."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))


def main():
    sys.path.insert(0, REPO)
    from gaussreg_amd.gs_io import PROPERTIES, read_gs_ply, write_gs_ply  # I/O helpers only
    sys.path.remove(REPO)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    for m in [k for k in sys.modules if k.startswith("geotransformer")]:
        del sys.modules[m]

    class _Prop:
        def __init__(self, name):
            self.name = name

    class _Elem:
        def __init__(self, rec):
            self.rec = rec
            self.properties = [_Prop(p) for p in PROPERTIES]

        def __getitem__(self, name):
            return self.rec[:, PROPERTIES.index(name)]

    class PlyData:
        def __init__(self, elements):
            self.elements = elements

        @staticmethod
        def read(path):
            return PlyData([_Elem(read_gs_ply(path))])

        def write(self, path):
            el = self.elements[0]
            write_gs_ply(path, np.stack([el[p] for p in PROPERTIES], 1).astype(np.float32))

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            return {p: elements[p] for p in PROPERTIES} if False else _StructElem(elements)

    class _StructElem:
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, name):
            return self.arr[name]

    stub = types.ModuleType("plyfile")
    stub.PlyData, stub.PlyElement = PlyData, PlyElement
    sys.modules["plyfile"] = stub
    sys.path.insert(0, "/root/reference")
    import gs_fusion as ref

    rng = np.random.default_rng(0)

    def cloud(n, centre):
        rec = np.zeros((n, 62), np.float32)
        rec[:, 0:3] = rng.normal(0, 1.0, (n, 3)) + centre
        rec[:, 6:54] = rng.normal(0, 0.3, (n, 48))
        rec[:, 54] = rng.normal(0, 2.0, n)
        rec[:, 55:58] = rng.normal(-4.0, 0.5, (n, 3))
        q = rng.normal(size=(n, 4))
        rec[:, 58:62] = q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.5, 2.0, (n, 1))
        return rec

    rec1, rec2 = cloud(700, [0, 0, 0]), cloud(600, [0.5, 0.2, -0.1])
    # non-zero input normals: the reference's save_ply writes zeros whatever the inputs held (gs_fusion.py:186-187)
    nrng = np.random.default_rng(77)
    rec1[:, 3:6] = nrng.normal(0, 1.0, (700, 3))
    rec2[:, 3:6] = nrng.normal(0, 1.0, (600, 3))
    # similarity transform: rotation * 1.37 + translation
    ax = np.array([0.3, -0.5, 0.8]); ax /= np.linalg.norm(ax); ang = 0.9
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    T = np.eye(4); T[:3, :3] = R * 1.37; T[:3, 3] = [1.5, -0.4, 0.3]
    with tempfile.TemporaryDirectory() as d:
        p1, p2, pt, po = [os.path.join(d, f) for f in ("a.ply", "b.ply", "t.npz", "o.ply")]
        write_gs_ply(p1, rec1); write_gs_ply(p2, rec2)
        np.savez(pt, estimated_transform=T)
        np.random.seed(0)
        ref.gaussian_fuse(p1, p2, pt, po)
        fused = read_gs_ply(po)
    np.savez_compressed(os.path.join(HERE, "gs_fusion.npz"), rec1=rec1, rec2=rec2, transform=T, fused=fused)
    print("gs_fusion.npz", os.path.getsize(os.path.join(HERE, "gs_fusion.npz")) // 1024, "KiB; fused", fused.shape)


if __name__ == "__main__":
    main()
