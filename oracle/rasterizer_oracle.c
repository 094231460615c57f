/* TEST INFRASTRUCTURE ONLY -- the product path never imports, links or calls this.
 *
 * PARITY UNPINNED: the 3DGS rasterizer is NOT part of /root/reference (SURVEY.md section 0 F3:
 * no diff_gaussian_rasterization source, submodule, pin or call site; the only trace is the
 * acknowledgement at README.md:150).  This file restates the PUBLISHED forward algorithm of
 * graphdeco-inria/diff-gaussian-rasterization (Kerbl et al. 2023; unpinned -- no version is
 * named anywhere in the reference) as summarised in SURVEY.md Appendix B, and is pinned only
 * by analytic known-answer tests (tests/test_oracle_rasterizer.py) and by the in-tree SH
 * convention it can be checked against (geotransformer/utils/graphics_utils.py:3-20 constants,
 * :34-77 polynomial -- same basis and signs as eval_sh there).
 *
 * Numerics contract (this is the repo's own spec; the HIP kernels implement the same sequence
 * of IEEE-754 binary32 operations, so the HIP image is compared BIT-EXACT with this file):
 *   - every `a*b+c` that is fused is written as fmaf(); everything else is separate mul/add
 *     (build with -ffp-contract=off);  sqrtf, '/', ceilf, rintf are correctly rounded;
 *   - exp() in the blend is exp_det() below (base-2 range reduction through the float's own bits + a degree-5
 *     polynomial, all fmaf: thirteen operations) -- within 2.4e-7 relative of exp where alpha can reach 1/255, and
 *     reproducible on any IEEE machine, CPU or GPU;
 *   - per-tile depth order = ascending (depth bits, Gaussian index) == a stable radix sort of
 *     (tile << 32 | depth_bits) keys emitted in Gaussian order (upstream's cub SortPairs).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* Deterministic expf for x <= 0 (clamped at -86: exp(-86) = 4.5e-38 is still a normal number).  Base-2 form, thirteen
 * basic operations (round 6; rounds 1 - 5 used the Cephes form, twenty):
 *   t  = x * log2(e)
 *   tm = t + 1.5 * 2^23                 the add rounds t to the nearest integer n (ties to even); n sits in tm's low bits
 *   nf = tm - 1.5 * 2^23                n as a float (exact)
 *   f  = fma(x, log2(e), -nf)           the reduced argument from the EXACT product: |f| <= 0.5 (+ 1 ulp of t)
 *   p  = 2^f, degree-5 polynomial with p(0) = 1, Horner in fmaf (minimax fit on [-0.5, 0.5]: 7e-8)
 *   result = bits(p) + (bits(tm) << 23) the integer add puts n into p's exponent field (n << 23 mod 2^32; -125 <= n <= 0)
 * Within 2.4e-7 relative of exp on [-6, 0] (where alpha can reach 1/255), 4.3e-7 on [-20, 0]; exp_det(0) == 1 exactly;
 * every operation is a correctly rounded IEEE-754 binary32 operation or integer arithmetic: the same bits on any machine. */
float oracle_exp_det(float x) {
  x = fmaxf(x, -86.0f);
  const float L2E = 1.44269504088896341f, MAGIC = 12582912.0f;
  const float t = x * L2E;
  const float tm = t + MAGIC;
  const float nf = tm - MAGIC;
  const float f = fmaf(x, L2E, -nf);
  float p = 1.3264815788716078e-3f;
  p = fmaf(p, f, 9.671512059867382e-3f);
  p = fmaf(p, f, 5.550733581185341e-2f);
  p = fmaf(p, f, 2.4022242426872253e-1f);
  p = fmaf(p, f, 6.931470036506653e-1f);
  p = fmaf(p, f, 1.0f);
  uint32_t pb, tb;
  memcpy(&pb, &p, 4);
  memcpy(&tb, &tm, 4);
  pb += tb << 23;
  float r;
  memcpy(&r, &pb, 4);
  return r;
}

static inline void xform4x3(const float* M, const float* p, float* o) {
  o[0] = fmaf(M[0], p[0], fmaf(M[4], p[1], fmaf(M[8], p[2], M[12])));
  o[1] = fmaf(M[1], p[0], fmaf(M[5], p[1], fmaf(M[9], p[2], M[13])));
  o[2] = fmaf(M[2], p[0], fmaf(M[6], p[1], fmaf(M[10], p[2], M[14])));
}
static inline void xform4x4(const float* M, const float* p, float* o) {
  o[0] = fmaf(M[0], p[0], fmaf(M[4], p[1], fmaf(M[8], p[2], M[12])));
  o[1] = fmaf(M[1], p[0], fmaf(M[5], p[1], fmaf(M[9], p[2], M[13])));
  o[2] = fmaf(M[2], p[0], fmaf(M[6], p[1], fmaf(M[10], p[2], M[14])));
  o[3] = fmaf(M[3], p[0], fmaf(M[7], p[1], fmaf(M[11], p[2], M[15])));
}

/* Sigma = R S^2 R^T from (scale*mod, quaternion wxyz, used un-normalised as upstream does). */
static void cov3d_from_scale_rot(const float* sc, float mod, const float* q, float* c6) {
  const float s0 = mod * sc[0], s1 = mod * sc[1], s2 = mod * sc[2];
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  float R[3][3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - r * z);
  R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);
  R[2][1] = 2.f * (y * z + r * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
  float M[3][3]; /* M[k][i] = s_k * R[i][k] */
  for (int i = 0; i < 3; ++i) {
    M[0][i] = s0 * R[i][0];
    M[1][i] = s1 * R[i][1];
    M[2][i] = s2 * R[i][2];
  }
#define SIG(i, j) fmaf(M[0][i], M[0][j], fmaf(M[1][i], M[1][j], M[2][i] * M[2][j]))
  c6[0] = SIG(0, 0);
  c6[1] = SIG(0, 1);
  c6[2] = SIG(0, 2);
  c6[3] = SIG(1, 1);
  c6[4] = SIG(1, 2);
  c6[5] = SIG(2, 2);
#undef SIG
}

static void cov2d(const float* t_in, float fx, float fy, float tanx, float tany, const float* c6,
                  const float* V, float* out3) {
  float t[3] = {t_in[0], t_in[1], t_in[2]};
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t[0] / t[2], tytz = t[1] / t[2];
  t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
  t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
  const float J00 = fx / t[2];
  const float J02 = -(fx * t[0]) / (t[2] * t[2]);
  const float J11 = fy / t[2];
  const float J12 = -(fy * t[1]) / (t[2] * t[2]);
  /* A = J * Vrot, Vrot(i,j) = V[j*4+i] (matrix passed transposed, read column-major) */
  float A0[3], A1[3];
  for (int j = 0; j < 3; ++j) {
    A0[j] = fmaf(J00, V[j * 4 + 0], J02 * V[j * 4 + 2]);
    A1[j] = fmaf(J11, V[j * 4 + 1], J12 * V[j * 4 + 2]);
  }
  const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float B0[3], B1[3];
  for (int k = 0; k < 3; ++k) {
    B0[k] = fmaf(S[k][0], A0[0], fmaf(S[k][1], A0[1], S[k][2] * A0[2]));
    B1[k] = fmaf(S[k][0], A1[0], fmaf(S[k][1], A1[1], S[k][2] * A1[2]));
  }
  out3[0] = fmaf(A0[0], B0[0], fmaf(A0[1], B0[1], A0[2] * B0[2])) + 0.3f;
  out3[1] = fmaf(A1[0], B0[0], fmaf(A1[1], B0[1], A1[2] * B0[2]));
  out3[2] = fmaf(A1[0], B1[0], fmaf(A1[1], B1[1], A1[2] * B1[2])) + 0.3f;
}

/* colour from SH (deg 0..3), coefficient layout (P, M, 3); basis identical to
 * geotransformer/utils/graphics_utils.py:34-77 */
static void sh_to_rgb(int deg, int M, const float* pos, const float* campos, const float* sh,
                      float* rgb) {
  float dx = pos[0] - campos[0], dy = pos[1] - campos[1], dz = pos[2] - campos[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx / len, y = dy / len, z = dz / len;
  (void)M;
  for (int c = 0; c < 3; ++c) {
#define S(k) sh[(k)*3 + c]
    float res = SH_C0 * S(0);
    if (deg > 0) {
      res = res - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) +
              SH_C2[2] * (2.0f * zz - xx - yy) * S(6) + SH_C2[3] * xz * S(7) +
              SH_C2[4] * (xx - yy) * S(8);
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
        }
      }
    }
#undef S
    res += 0.5f;
    rgb[c] = fmaxf(res, 0.0f);
  }
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static void get_rect(float px, float py, int radius, int gx, int gy, int* rmin, int* rmax) {
  const float r = (float)radius;
  rmin[0] = imin(gx, imax(0, (int)((px - r) / (float)TILE)));
  rmin[1] = imin(gy, imax(0, (int)((py - r) / (float)TILE)));
  rmax[0] = imin(gx, imax(0, (int)((px + r + (float)(TILE - 1)) / (float)TILE)));
  rmax[1] = imin(gy, imax(0, (int)((py + r + (float)(TILE - 1)) / (float)TILE)));
}

typedef struct {
  uint32_t depth_bits;
  int32_t id;
} ent_t;

/* stable merge sort by depth_bits (ties keep emission order == ascending id) */
static void msort(ent_t* a, ent_t* tmp, int64_t n) {
  if (n < 2) return;
  int64_t h = n / 2;
  msort(a, tmp, h);
  msort(a + h, tmp, n - h);
  int64_t i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (a[j].depth_bits < a[i].depth_bits) ? a[j++] : a[i++];
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, sizeof(ent_t) * (size_t)n);
}

/* Per-Gaussian preprocess; also exported on its own so tests can compare the HIP
 * preprocess outputs field by field.  Arrays sized P (xy: 2P, conic_opacity: 4P, rgb: 3P). */
void oracle_raster_preprocess(int P, int D, int M, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities,
                              const float* scales, float scale_modifier, const float* rotations,
                              const float* cov3D_precomp, const float* viewmatrix,
                              const float* projmatrix, const float* cam_pos, int W, int H,
                              float tan_fovx, float tan_fovy, int* radii, float* xy, float* depths,
                              float* conic_opacity, float* rgb, int* tiles_touched) {
  const float fx = (float)W / (2.0f * tan_fovx), fy = (float)H / (2.0f * tan_fovy);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  for (int i = 0; i < P; ++i) {
    radii[i] = 0;
    tiles_touched[i] = 0;
    xy[2 * i] = xy[2 * i + 1] = 0.f;
    depths[i] = 0.f;
    for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0.f;
    for (int k = 0; k < 3; ++k) rgb[3 * i + k] = 0.f;
    const float* p = means3D + 3 * i;
    float pv[3];
    xform4x3(viewmatrix, p, pv);
    if (pv[2] <= 0.2f) continue; /* near cull */
    float ph[4];
    xform4x4(projmatrix, p, ph);
    const float pw = 1.0f / (ph[3] + 0.0000001f);
    const float pproj[2] = {ph[0] * pw, ph[1] * pw};
    float c6[6];
    if (cov3D_precomp)
      memcpy(c6, cov3D_precomp + 6 * i, sizeof(c6));
    else
      cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rotations + 4 * i, c6);
    float cv[3];
    cov2d(pv, fx, fy, tan_fovx, tan_fovy, c6, viewmatrix, cv);
    const float det = cv[0] * cv[2] - cv[1] * cv[1];
    if (det == 0.0f) continue;
    const float det_inv = 1.f / det;
    const float conic[3] = {cv[2] * det_inv, -cv[1] * det_inv, cv[0] * det_inv};
    const float mid = 0.5f * (cv[0] + cv[2]);
    const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float l1 = mid + sq, l2 = mid - sq;
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
    const float px = ((pproj[0] + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float py = ((pproj[1] + 1.0f) * (float)H - 1.0f) * 0.5f;
    int rmin[2], rmax[2];
    get_rect(px, py, (int)my_radius, gx, gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
    if (colors_precomp)
      memcpy(rgb + 3 * i, colors_precomp + 3 * i, 3 * sizeof(float));
    else
      sh_to_rgb(D, M, p, cam_pos, shs + (size_t)i * M * 3, rgb + 3 * i);
    depths[i] = pv[2];
    radii[i] = (int)my_radius;
    xy[2 * i] = px;
    xy[2 * i + 1] = py;
    conic_opacity[4 * i + 0] = conic[0];
    conic_opacity[4 * i + 1] = conic[1];
    conic_opacity[4 * i + 2] = conic[2];
    conic_opacity[4 * i + 3] = opacities[i];
    tiles_touched[i] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
  }
}

/* Full forward.  out_color: (3,H,W); radii: (P).  Returns the number of (tile, Gaussian)
 * instances R ("num_rendered"), or <0 on allocation failure. */
int64_t oracle_rasterize_forward(int P, int D, int M, const float* background, int W, int H,
                                 const float* means3D, const float* shs,
                                 const float* colors_precomp, const float* opacities,
                                 const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp,
                                 const float* viewmatrix, const float* projmatrix,
                                 const float* cam_pos, float tan_fovx, float tan_fovy,
                                 float* out_color, int* radii) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int ntiles = gx * gy;
  float* xy = (float*)malloc(sizeof(float) * 2 * (size_t)(P + 1));
  float* depths = (float*)malloc(sizeof(float) * (size_t)(P + 1));
  float* co = (float*)malloc(sizeof(float) * 4 * (size_t)(P + 1));
  float* rgb = (float*)malloc(sizeof(float) * 3 * (size_t)(P + 1));
  int* tt = (int*)malloc(sizeof(int) * (size_t)(P + 1));
  int64_t* tcount = (int64_t*)calloc((size_t)ntiles + 1, sizeof(int64_t));
  if (!xy || !depths || !co || !rgb || !tt || !tcount) return -1;
  oracle_raster_preprocess(P, D, M, means3D, shs, colors_precomp, opacities, scales,
                           scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                           cam_pos, W, H, tan_fovx, tan_fovy, radii, xy, depths, co, rgb, tt);
  /* binning: per-tile lists in Gaussian order, then stable sort by depth */
  for (int i = 0; i < P; ++i) {
    if (radii[i] <= 0) continue;
    int rmin[2], rmax[2];
    get_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
    for (int y = rmin[1]; y < rmax[1]; ++y)
      for (int x = rmin[0]; x < rmax[0]; ++x) tcount[y * gx + x + 1]++;
  }
  for (int t = 0; t < ntiles; ++t) tcount[t + 1] += tcount[t];
  const int64_t R = tcount[ntiles];
  ent_t* ents = (ent_t*)malloc(sizeof(ent_t) * (size_t)(R + 1));
  ent_t* tmp = (ent_t*)malloc(sizeof(ent_t) * (size_t)(R + 1));
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ntiles + 1));
  if (!ents || !tmp || !cur) return -1;
  memcpy(cur, tcount, sizeof(int64_t) * (size_t)ntiles);
  for (int i = 0; i < P; ++i) {
    if (radii[i] <= 0) continue;
    int rmin[2], rmax[2];
    get_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
    uint32_t db;
    memcpy(&db, &depths[i], 4);
    for (int y = rmin[1]; y < rmax[1]; ++y)
      for (int x = rmin[0]; x < rmax[0]; ++x) {
        ent_t e = {db, i};
        ents[cur[y * gx + x]++] = e;
      }
  }
  for (int t = 0; t < ntiles; ++t) msort(ents + tcount[t], tmp, tcount[t + 1] - tcount[t]);
  /* blend, one pixel at a time, front to back */
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const int t = (py / TILE) * gx + (px / TILE);
      const float pfx = (float)px, pfy = (float)py;
      float T = 1.0f, C[3] = {0.f, 0.f, 0.f};
      for (int64_t k = tcount[t]; k < tcount[t + 1]; ++k) {
        const int id = ents[k].id;
        const float dx = xy[2 * id] - pfx, dy = xy[2 * id + 1] - pfy;
        const float* c = co + 4 * id;
        /* power = -0.5*(cx*dx*dx + cz*dy*dy) - cy*dx*dy, fused as written */
        const float q = fmaf(c[0] * dx, dx, (c[2] * dy) * dy);
        const float power = fmaf(-0.5f, q, -((c[1] * dx) * dy));
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, c[3] * oracle_exp_det(power));
        if (alpha < 1.0f / 255.0f) continue;
        const float test_T = T * (1.0f - alpha);
        if (test_T < 0.0001f) break;
        const float w = alpha * T;
        C[0] = fmaf(rgb[3 * id + 0], w, C[0]);
        C[1] = fmaf(rgb[3 * id + 1], w, C[1]);
        C[2] = fmaf(rgb[3 * id + 2], w, C[2]);
        T = test_T;
      }
      for (int ch = 0; ch < 3; ++ch)
        out_color[(size_t)ch * H * W + (size_t)py * W + px] = fmaf(T, background[ch], C[ch]);
    }
  free(xy);
  free(depths);
  free(co);
  free(rgb);
  free(tt);
  free(tcount);
  free(ents);
  free(tmp);
  free(cur);
  return R;
}

/* markVisible: same near-plane test as the preprocess cull */
void oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present) {
  for (int i = 0; i < P; ++i) {
    float pv[3];
    xform4x3(viewmatrix, means3D + 3 * i, pv);
    present[i] = pv[2] > 0.2f ? 1 : 0;
  }
}
