#!/usr/bin/env python3
"""Golden vector for the SH basis/sign convention, FROM THE REFERENCE's eval_sh
(geotransformer/utils/graphics_utils.py:34-77).  Runs only where /root/reference is mounted.
The reference file is loaded as a module by path (it imports nothing but torch-free code)."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location(
        "ref_graphics_utils", "/root/reference/geotransformer/utils/graphics_utils.py")
    gu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gu)
    rng = np.random.default_rng(7)
    P = 400
    means = ((rng.random((P, 3)) - 0.5) * [3.0, 2.0, 2.0] + [0, 0, 3.0]).astype(np.float32)
    campos = np.float32([0.2, -0.1, 0.05])
    shs = rng.normal(0, 0.3, (P, 16, 3)).astype(np.float32)   # (P, K, 3) rasterizer layout
    dirs = means - campos
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {"means": means, "campos": campos, "shs": shs}
    sh_ref = np.transpose(shs, (0, 2, 1))                      # eval_sh wants (..., C, K)
    for deg in range(4):
        out[f"eval_deg{deg}"] = np.asarray(gu.eval_sh(deg, sh_ref[..., :(deg + 1) ** 2], dirs), np.float32)
    np.savez_compressed(os.path.join(HERE, "sh_eval.npz"), **out)
    print("sh_eval.npz", os.path.getsize(os.path.join(HERE, "sh_eval.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
