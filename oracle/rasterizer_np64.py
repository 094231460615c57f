"""An INDEPENDENT second opinion on the rasterizer: the published 3DGS forward pass (Kerbl et al. 2023)
written straight from the paper-level description in float64 NumPy.  TEST INFRASTRUCTURE ONLY.

It deliberately shares nothing with oracle/rasterizer_oracle.c (which defines the bit-exact fp32 operation sequence the
HIP kernels follow): no fmaf, no polynomial exp (numpy / libm exp), float64 everywhere, covariance built with matrix
products, SH evaluated through an explicit basis vector, compositing vectorised per Gaussian over its tile rectangle.
What it keeps are the RULES of the algorithm, because they decide which Gaussians a pixel sees:
  near cull z <= 0.2, +0.3 px^2 dilation, radius = ceil(3 sqrt(lambda_max)), 16 x 16 tiles and the tile rectangle of
  (centre +- radius), depth order (ties by index), skip power > 0, alpha = min(0.99, opacity * exp(power)), skip
  alpha < 1/255, stop BEFORE blending once T (1 - alpha) < 1e-4, final colour C + T * background.
Agreement of the fp32 pipeline with this renderer bounds how far the chosen fp32 sequence sits from the real-number
algorithm (threshold decisions made in fp32 vs fp64 can differ on isolated pixels; the tests report and bound those).
"""
import numpy as np

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def sh_basis(dirs, degree):
    """Real SH basis (the sign conventions of the 3DGS code base), (N, (degree+1)^2)."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    cols = [np.full_like(x, SH_C0)]
    if degree > 0:
        cols += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
        if degree > 2:
            cols += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                     SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                     SH_C3[6] * x * (xx - 3 * yy)]
    return np.stack(cols, axis=1)


def render(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, viewmatrix,
           projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0, scale_modifier=1.0):
    """-> (image (3, H, W) float64, radii (P,) int64, stats dict).  Matrices arrive transposed, as the Python API passes them."""
    f8 = np.float64
    p = np.asarray(means3D, f8).reshape(-1, 3)
    P = p.shape[0]
    op = np.asarray(opacities, f8).reshape(-1)
    V = np.asarray(viewmatrix, f8).reshape(4, 4).T     # world -> view, column-vector convention
    PM = np.asarray(projmatrix, f8).reshape(4, 4).T    # full projection
    cam = np.asarray(campos, f8).reshape(3)
    ph = np.concatenate([p, np.ones((P, 1))], 1)
    pv = ph @ V.T
    hom = ph @ PM.T
    ndc = hom[:, :2] / (hom[:, 3:4] + 1e-7)
    visible = pv[:, 2] > 0.2
    # ---- 3-D covariance
    if cov3D_precomp is not None:
        c = np.asarray(cov3D_precomp, f8).reshape(-1, 6)
        S3 = np.stack([np.stack([c[:, 0], c[:, 1], c[:, 2]], 1), np.stack([c[:, 1], c[:, 3], c[:, 4]], 1),
                       np.stack([c[:, 2], c[:, 4], c[:, 5]], 1)], 1)
    else:
        q = np.asarray(rotations, f8).reshape(-1, 4)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
                      np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
                      np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
        s = np.asarray(scales, f8).reshape(-1, 3) * float(scale_modifier)
        M = R * s[:, None, :]                      # R diag(s)
        S3 = M @ np.transpose(M, (0, 2, 1))        # R S S^T R^T
    # ---- 2-D covariance (EWA splatting)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = np.where(visible, pv[:, 2], 1.0)
    tx = np.clip(pv[:, 0] / tz, -1.3 * tanfovx, 1.3 * tanfovx) * tz
    ty = np.clip(pv[:, 1] / tz, -1.3 * tanfovy, 1.3 * tanfovy) * tz
    J = np.zeros((P, 2, 3))
    J[:, 0, 0], J[:, 0, 2] = fx / tz, -fx * tx / (tz * tz)
    J[:, 1, 1], J[:, 1, 2] = fy / tz, -fy * ty / (tz * tz)
    T = J @ V[:3, :3][None]
    cov = T @ S3 @ np.transpose(T, (0, 2, 1))
    a, b, c2 = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c2 - b * b
    ok = visible & (det != 0)
    det_s = np.where(ok, det, 1.0)
    conic = np.stack([c2 / det_s, -b / det_s, a / det_s], 1)
    mid = 0.5 * (a + c2)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    radius = np.ceil(3.0 * np.sqrt(lam)).astype(np.int64)
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16

    def tile_lo(v, g):
        return np.clip(np.trunc(v / 16.0).astype(np.int64), 0, g)

    with np.errstate(invalid="ignore", over="ignore"):
        x0, x1 = tile_lo(np.where(ok, px - radius, 0.0), gx), tile_lo(np.where(ok, px + radius + 15, 0.0), gx)
        y0, y1 = tile_lo(np.where(ok, py - radius, 0.0), gy), tile_lo(np.where(ok, py + radius + 15, 0.0), gy)
    ok &= (x1 > x0) & (y1 > y0)
    radii = np.where(ok, radius, 0)
    # ---- colours
    if colors_precomp is not None:
        rgb = np.asarray(colors_precomp, f8).reshape(-1, 3)
    else:
        sh = np.asarray(shs, f8).reshape(P, -1, 3)
        d = p - cam[None]
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        B = sh_basis(d, int(sh_degree))
        rgb = np.maximum(np.einsum('nk,nkc->nc', B, sh[:, :B.shape[1], :]) + 0.5, 0.0)
    # ---- front-to-back compositing, one Gaussian at a time over its tile rectangle
    order = np.argsort(np.where(ok, pv[:, 2], np.inf), kind="stable")
    order = order[: int(ok.sum())]
    Tm = np.ones((H, W))
    C = np.zeros((3, H, W))
    done = np.zeros((H, W), bool)
    n_pairs = 0
    for i in order:
        xa, xb = int(x0[i]) * 16, min(int(x1[i]) * 16, W)
        ya, yb = int(y0[i]) * 16, min(int(y1[i]) * 16, H)
        sub_done = done[ya:yb, xa:xb]
        if sub_done.all():
            continue
        X = px[i] - np.arange(xa, xb, dtype=f8)[None, :]
        Y = py[i] - np.arange(ya, yb, dtype=f8)[:, None]
        power = -0.5 * (conic[i, 0] * X * X + conic[i, 2] * Y * Y) - conic[i, 1] * X * Y
        alpha = np.minimum(0.99, op[i] * np.exp(np.minimum(power, 0.0)))
        hit = (power <= 0.0) & (alpha >= 1.0 / 255.0) & ~sub_done
        if not hit.any():
            continue
        Tsub = Tm[ya:yb, xa:xb]
        test_T = Tsub * (1.0 - alpha)
        stop = hit & (test_T < 1e-4)
        blend = hit & ~stop
        w = np.where(blend, alpha * Tsub, 0.0)
        C[:, ya:yb, xa:xb] += rgb[i][:, None, None] * w[None]
        Tm[ya:yb, xa:xb] = np.where(blend, test_T, Tsub)
        done[ya:yb, xa:xb] |= stop
        n_pairs += int(blend.sum())
    img = C + Tm[None] * np.asarray(bg, f8).reshape(3, 1, 1)
    return img, radii, {"visible": int(ok.sum()), "blended_pairs": n_pairs}
