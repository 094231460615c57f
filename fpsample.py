"""Drop-in for the one `fpsample` entry point GaussReg uses (demo.py:46, test.py:46, dataset.py:127):
`fpsample.bucket_fps_kdline_sampling(points, n, h=9)`.  Exact FPS on the GPU (gaussreg_amd/csrc/fps.hip);
`h` (kd-tree height of the CPU algorithm) is accepted and ignored.  Parity unpinned: fpsample is not vendored
by the reference and draws its start index at random; pass start_idx for a reproducible result (default 0)."""
import numpy as np

from gaussreg_amd.registration import farthest_point_sampling


def bucket_fps_kdline_sampling(pc, n_samples, h=None, start_idx=None):
    pc = np.ascontiguousarray(pc, dtype=np.float32) if not hasattr(pc, "is_cuda") else pc
    if pc.ndim != 2 or pc.shape[1] != 3:
        raise ValueError("pc must be (N, 3)")
    if n_samples > pc.shape[0]:
        raise ValueError("n_samples must be <= number of points")
    idx = farthest_point_sampling(pc, [pc.shape[0]], [int(n_samples)],
                                  None if start_idx is None else [int(start_idx)])[0]
    return idx.cpu().numpy().astype(np.uint64)


bucket_fps_kdtree_sampling = bucket_fps_kdline_sampling
fps_sampling = bucket_fps_kdline_sampling
