"""TEST INFRASTRUCTURE ONLY.  NumPy restatement of gs_fusion.py:231-262 (gaussian_fuse) on (N,62) vertex
records, pinned against tests/golden/gs_fusion.npz (produced by the reference's own gaussian_fuse)."""
import numpy as np

C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _basis(d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    return (np.stack([-C1 * y, C1 * z, -C1 * x], 1),
            np.stack([C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)], 1),
            np.stack([C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                      C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                      C3[6] * x * (xx - 3 * yy)], 1))


def quat_to_mat(q):
    q = q.astype(np.float32)
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    two_s = np.float32(2.0) / (q * q).sum(-1, dtype=np.float32)
    o = np.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                  two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                  two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(-1, 3, 3).astype(np.float32)


def mat_to_quat(m):
    m = m.astype(np.float32)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[:, a, b] for a in range(3) for b in range(3)]
    qa = np.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1)
    qa = np.sqrt(np.maximum(qa, 0)).astype(np.float32)
    cand = np.stack([np.stack([qa[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                     np.stack([m21 - m12, qa[:, 1] ** 2, m10 + m01, m02 + m20], -1),
                     np.stack([m02 - m20, m10 + m01, qa[:, 2] ** 2, m12 + m21], -1),
                     np.stack([m10 - m01, m20 + m02, m21 + m12, qa[:, 3] ** 2], -1)], -2)
    cand = cand / (2.0 * np.maximum(qa[..., None], np.float32(0.1)))
    return cand[np.arange(m.shape[0]), qa.argmax(-1)].astype(np.float32)


def gaussian_fuse(rec1, rec2, T):
    rec1, rec2 = np.asarray(rec1, np.float32), np.asarray(rec2, np.float32)
    rot = T[:3, :3]
    scale = (rot @ rot.T)[0, 0] ** 0.5
    rot = rot / scale
    xyz2 = rec2[:, 0:3] @ rot.T * scale + T[None, :3, 3]
    out2 = rec2.astype(np.float64).copy()
    out2[:, 0:3] = xyz2
    out2[:, 55:58] = rec2[:, 55:58] + np.log(scale) if scale != 1.0 else rec2[:, 55:58]
    out2[:, 58:62] = mat_to_quat(np.matmul(rot[None].astype(np.float32), quat_to_mat(rec2[:, 58:62])))
    d = np.random.default_rng(5).normal(size=(40, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = _basis(d), _basis(d @ rot.T)
    sh = rec2[:, 9:54].reshape(-1, 3, 15).astype(np.float64)
    for (lo, hi), ai, bi in zip(((0, 3), (3, 8), (8, 15)), a, b):
        sh[:, :, lo:hi] = sh[:, :, lo:hi] @ (np.linalg.pinv(ai) @ bi)
    out2[:, 9:54] = sh.reshape(-1, 45)
    c1 = rec1[:, 0:3].mean(0)
    c2 = xyz2.mean(0)
    k1 = np.linalg.norm(rec1[:, 0:3] - c1, axis=1) < np.linalg.norm(rec1[:, 0:3] - c2, axis=1)
    k2 = np.linalg.norm(xyz2 - c2, axis=1) < np.linalg.norm(xyz2 - c1, axis=1)
    fused = np.concatenate([rec1[k1], out2[k2].astype(np.float32)], 0)
    fused[:, 3:6] = 0.0  # save_ply writes zero normals whatever the inputs held (gs_fusion.py:186-187)
    return fused
