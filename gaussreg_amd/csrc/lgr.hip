// LocalGlobalRegistration.local_to_global_registration on the GPU, no host round trips.
//
// Replaces  geotransformer/modules/geotransformer/local_global_registration.py:135-193  and the
// weighted Kabsch it calls, geotransformer/modules/registration/procrustes.py:6-82, which ships every 3x3
// covariance to the CPU for torch.svd (procrustes.py:59) -- 1 + num_refinement_steps GPU->CPU->GPU syncs
// per forward in the reference.  Input: the dense correspondences already gathered in torch.nonzero order
// (gr_corr_gather) plus the per-patch offsets / counts gr_corr_matrix left in the workspace.
//   local     one workgroup per patch correspondence with >= correspondence_threshold matches:
//             weighted centroids + 3x3 covariance by block reduction, optimal rotation in fp64
//   verify    one workgroup per hypothesis: inliers (residual < acceptance_radius) over ALL correspondences
//   refine    one workgroup: best hypothesis -> mask -> weighted Kabsch, num_refinement_steps times
// Rotation: the proper rotation maximising tr(R H) -- Horn's unit-quaternion form (largest eigenvector of a
// symmetric 4x4 built from H, cyclic Jacobi in fp64); identical to V diag(1,1,det(VU^T)) U^T of
// procrustes.py:59-64 whenever that is unique.
#include "common.hpp"

namespace gr {
namespace {

constexpr int LG_T = 256;

// largest-eigenvalue eigenvector of symmetric 4x4 A (row-major), cyclic Jacobi
__device__ void horn_rotation(const double* H /*3x3: H[a][b] = sum w src_a ref_b*/, double* R /*3x3*/) {
  const double Sxx = H[0], Sxy = H[1], Sxz = H[2], Syx = H[3], Syy = H[4], Syz = H[5], Szx = H[6], Szy = H[7], Szz = H[8];
  double A[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                    {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                    {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                    {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  // stop when the off-diagonal mass is below fp64 resolution of the matrix (1e-15 of its Frobenius norm): waiting for it
  // to underflow costs four more sweeps of serial fp64 work in the one thread that runs this
  double fro2 = 0.0;
  for (int p = 0; p < 4; ++p)
    for (int q = 0; q < 4; ++q) fro2 += A[p][q] * A[p][q];
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
    if (off <= 1e-30 * fro2) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (A[k][k] > A[best][best]) best = k;
  double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
  const double nrm = sqrt(w * w + x * x + y * y + z * z);
  w /= nrm; x /= nrm; y /= nrm; z /= nrm;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// block-wide sum of NV doubles per thread -> result in sh[0..NV) (valid for all threads after return)
template <int NV, int T>
__device__ void block_sum(double* v, double* sh /* [NV][T/64] + NV */) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k] = wave_sum_f64_dpp(v[k]);
  }
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  if (lane == 0)
    for (int k = 0; k < NV; ++k) sh[k * (T / WAVE) + w] = v[k];
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0;
    for (int i = 0; i < T / WAVE; ++i) s += sh[threadIdx.x * (T / WAVE) + i];
    sh[NV * (T / WAVE) + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = sh[NV * (T / WAVE) + k];
}

// weighted Kabsch over correspondences [a, b) with weight(i); thread 0 ends up with (R, t) in T[12]
// procrustes.py:41-66: w = w / (sum w + eps); centroids; H = sum w (src - cs)(ref - cr)^T
template <int NT, typename WF>
__device__ void block_procrustes(const float* __restrict__ src, const float* __restrict__ ref, int a, int b, WF weight,
                                 float eps, double* sh, float* T /* shared [12] */) {
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};  // sum w, sum w*src (3), sum w*ref (3)
  for (int i = a + threadIdx.x; i < b; i += NT) {
    const double w = weight(i);
    acc[0] += w;
    for (int k = 0; k < 3; ++k) {
      acc[1 + k] += w * src[3 * (int64_t)i + k];
      acc[4 + k] += w * ref[3 * (int64_t)i + k];
    }
  }
  block_sum<7, NT>(acc, sh);
  const double wn = 1.0 / (acc[0] + (double)eps);
  const double cs[3] = {acc[1] * wn, acc[2] * wn, acc[3] * wn}, cr[3] = {acc[4] * wn, acc[5] * wn, acc[6] * wn};
  double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = a + threadIdx.x; i < b; i += NT) {
    const double w = weight(i) * wn;
    double s[3], r[3];
    for (int k = 0; k < 3; ++k) {
      s[k] = (double)src[3 * (int64_t)i + k] - cs[k];
      r[k] = (double)ref[3 * (int64_t)i + k] - cr[k];
    }
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) h[p * 3 + q] += s[p] * w * r[q];
  }
  block_sum<9, NT>(h, sh);
  if (threadIdx.x == 0) {
    double R[9];
    horn_rotation(h, R);
    for (int k = 0; k < 9; ++k) T[k] = (float)R[k];
    for (int r = 0; r < 3; ++r) T[9 + r] = (float)(cr[r] - (R[r * 3] * cs[0] + R[r * 3 + 1] * cs[1] + R[r * 3 + 2] * cs[2]));
  }
  __syncthreads();
}

__device__ __forceinline__ bool inlier(const float* T, const float* src, const float* ref, int64_t i, float radius) {
  const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
  // apply_transform: p R^T + t  (ops/transformation.py:38), then torch.linalg.norm
  const float ax = (sx * T[0] + sy * T[1] + sz * T[2]) + T[9];
  const float ay = (sx * T[3] + sy * T[4] + sz * T[5]) + T[10];
  const float az = (sx * T[6] + sy * T[7] + sz * T[8]) + T[11];
  const float dx = ref[3 * i] - ax, dy = ref[3 * i + 1] - ay, dz = ref[3 * i + 2] - az;
  return sqrtf((dx * dx + dy * dy) + dz * dz) < radius;
}

// per patch: local transform (or valid = 0)
__global__ __launch_bounds__(LG_T) void lgr_local_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                         const float* __restrict__ scores,
                                                         const int32_t* __restrict__ counts,
                                                         const int32_t* __restrict__ offsets, int min_corr,
                                                         float* __restrict__ transforms /* [B][12] */,
                                                         int32_t* __restrict__ valid) {
  __shared__ double sh[9 * (LG_T / WAVE) + 16];
  __shared__ float T[12];
  const int p = blockIdx.x;
  const int n = counts[p];
  if (n < min_corr) {
    if (threadIdx.x == 0) valid[p] = 0;
    return;
  }
  const int a = offsets[p];
  block_procrustes<LG_T>(src, ref, a, a + n, [&](int i) { return (double)fmaxf(scores[i], 0.0f); }, 1e-5f, sh, T);
  if (threadIdx.x < 12) transforms[p * 12 + threadIdx.x] = T[threadIdx.x];
  if (threadIdx.x == 0) valid[p] = 1;
}

// per hypothesis: number of inliers over all C correspondences
// (segmented form: seg_patch_off lists the first patch of each scene pair, row_off the first correspondence of every patch
//  -- the hypothesis of patch p is scored on the correspondences of p's own pair)
__global__ __launch_bounds__(LG_T) void lgr_verify_kernel(const float* __restrict__ src, const float* __restrict__ ref, int C,
                                                          const float* __restrict__ transforms,
                                                          const int32_t* __restrict__ valid, float radius,
                                                          int32_t* __restrict__ inliers,
                                                          const int32_t* __restrict__ seg_patch_off, int nseg,
                                                          const int32_t* __restrict__ row_off) {
  __shared__ int wsum[LG_T / WAVE];
  __shared__ float T[12];
  const int p = blockIdx.x;
  if (seg_patch_off) {
    const int sgm = find_batch(seg_patch_off, nseg, p);
    const int a = row_off[seg_patch_off[sgm]];
    C = row_off[seg_patch_off[sgm + 1]] - a;
    src += 3 * (int64_t)a;
    ref += 3 * (int64_t)a;
  }
  if (!valid[p]) {
    if (threadIdx.x == 0) inliers[p] = -1;
    return;
  }
  if (threadIdx.x < 12) T[threadIdx.x] = transforms[p * 12 + threadIdx.x];
  __syncthreads();
  int n = 0;
  for (int i = threadIdx.x; i < C; i += LG_T) n += inlier(T, src, ref, i, radius) ? 1 : 0;
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) n += __shfl_xor(n, d, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0) wsum[threadIdx.x / WAVE] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < LG_T / WAVE; ++i) t += wsum[i];
    inliers[p] = t;
  }
}

// The same counts in stack mode, one THREAD per hypothesis: a workgroup takes one slice of one scene pair's correspondences
// (staged in LDS as coordinate planes, read as broadcasts, two per packed fp32 instruction) and runs all the pair's patch
// hypotheses over it -- 256 at a time, transforms in registers -- instead of one workgroup per hypothesis streaming all of
// the pair's correspondences from L2 (64 pairs x 256 hypotheses x 12 000 correspondences: 16 384 workgroups x 288 KB =
// 4.7 GB of L2 reads, 0.52 ms).  Same arithmetic as inlier() above; `sqrtf(d2) < radius` is evaluated as `d2 < x0` with x0 the
// smallest float whose sqrtf reaches the radius (found once per workgroup with the same sqrtf: it is monotone).
// inliers[] must be zero on entry; the slices add their counts (integers: any order), slice 0 marks the invalid patches.
constexpr int LGV_CHUNK = 1024, LGV_SLICES = 16;
typedef float lg_f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(LG_T) void lgr_verify_wide_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                               const float* __restrict__ transforms,
                                                               const int32_t* __restrict__ valid, float radius,
                                                               int32_t* __restrict__ inliers,
                                                               const int32_t* __restrict__ seg_patch_off,
                                                               const int32_t* __restrict__ row_off) {
  __shared__ __attribute__((aligned(16))) float s_c[6][LGV_CHUNK];  // planes: src x, y, z, ref x, y, z
  __shared__ float s_x0;
  const int pa = seg_patch_off[blockIdx.y], pe = seg_patch_off[blockIdx.y + 1];
  const int a = row_off[pa], C = row_off[pe] - a;
  src += 3 * (int64_t)a;
  ref += 3 * (int64_t)a;
  // this workgroup's slice of the pair's correspondences: whole pairs of rows
  const int per = ((C + LGV_SLICES - 1) / LGV_SLICES + 1) & ~1;
  const int c_lo = min((int)blockIdx.x * per, C), c_hi = min(c_lo + per, C);
  if (threadIdx.x == 0) {
    unsigned lo = 0u, hi = 0x7f800000u;  // smallest bit pattern b in [lo, hi] with sqrtf(float(b)) >= radius (hi: sqrtf(inf) = inf)
    if (!(radius == radius)) lo = hi = 0u;  // NaN radius: nothing is an inlier (d2 < 0 never holds)
    else if (!(sqrtf(__uint_as_float(hi)) >= radius)) lo = hi;
    while (lo < hi) {
      const unsigned mid = lo + (hi - lo) / 2u;
      if (sqrtf(__uint_as_float(mid)) >= radius) hi = mid;
      else lo = mid + 1u;
    }
    s_x0 = __uint_as_float(lo);
  }
  if (blockIdx.x == 0)
    for (int p = pa + (int)threadIdx.x; p < pe; p += LG_T)
      if (!valid[p]) inliers[p] = -1;
  __syncthreads();
  const float x0 = s_x0;
  for (int pb = pa; pb < pe; pb += LG_T) {
    const int p = pb + (int)threadIdx.x;
    const bool ok = p < pe && valid[p] != 0;
    lg_f2 T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float t = ok ? transforms[(int64_t)p * 12 + k] : 0.f;
      T[k] = lg_f2{t, t};
    }
    int cnt = 0;
    for (int c0 = c_lo; c0 < c_hi; c0 += LGV_CHUNK) {
      const int nn = min(LGV_CHUNK, c_hi - c0), np = (nn + 1) & ~1;
      __syncthreads();
      for (int e = threadIdx.x; e < nn * 3; e += LG_T) {
        const int i = e / 3, q = e - 3 * i;
        s_c[q][i] = src[3 * (int64_t)c0 + e];
        s_c[3 + q][i] = ref[3 * (int64_t)c0 + e];
      }
      if (threadIdx.x == 0 && np > nn) {  // the odd row's partner: an inlier of nothing
        s_c[0][nn] = s_c[1][nn] = s_c[2][nn] = 0.f;
        s_c[3][nn] = s_c[4][nn] = s_c[5][nn] = INFINITY;
      }
      __syncthreads();
      if (ok) {
#pragma unroll 4
        for (int i = 0; i < np; i += 2) {
          const lg_f2 sx = *reinterpret_cast<const lg_f2*>(&s_c[0][i]), sy = *reinterpret_cast<const lg_f2*>(&s_c[1][i]);
          const lg_f2 sz = *reinterpret_cast<const lg_f2*>(&s_c[2][i]);
          const lg_f2 ax = (sx * T[0] + sy * T[1] + sz * T[2]) + T[9];
          const lg_f2 ay = (sx * T[3] + sy * T[4] + sz * T[5]) + T[10];
          const lg_f2 az = (sx * T[6] + sy * T[7] + sz * T[8]) + T[11];
          const lg_f2 dx = *reinterpret_cast<const lg_f2*>(&s_c[3][i]) - ax, dy = *reinterpret_cast<const lg_f2*>(&s_c[4][i]) - ay;
          const lg_f2 dz = *reinterpret_cast<const lg_f2*>(&s_c[5][i]) - az;
          const lg_f2 d2 = (dx * dx + dy * dy) + dz * dz;
          cnt += (d2.x < x0 ? 1 : 0) + (d2.y < x0 ? 1 : 0);
        }
      }
    }
    if (ok && cnt) atomicAdd(&inliers[p], cnt);
  }
}

// best hypothesis + global refinement (local_global_registration.py:171-192): one workgroup, so a wide one (the two passes
// of every refinement step stream all correspondences)
constexpr int LG_RT = 1024;
__global__ __launch_bounds__(LG_RT) void lgr_refine_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                          const float* __restrict__ scores, int C, int B,
                                                          const float* __restrict__ transforms,
                                                          const int32_t* __restrict__ inliers, float radius, int steps,
                                                          float* __restrict__ out_transform /* 4x4 row-major */,
                                                          const int32_t* __restrict__ seg_patch_off,
                                                          const int32_t* __restrict__ row_off,
                                                          int32_t* __restrict__ out_seg_rows) {
  __shared__ double sh[9 * (LG_RT / WAVE) + 16];
  __shared__ float T[12];
  __shared__ int best_sh;
  if (seg_patch_off) {  // workgroup = scene pair: its patches [pa, pe), its correspondences [a, a + C)
    const int pa = seg_patch_off[blockIdx.x], pe = seg_patch_off[blockIdx.x + 1];
    const int a = row_off[pa];
    C = row_off[pe] - a;
    B = pe - pa;
    src += 3 * (int64_t)a;
    ref += 3 * (int64_t)a;
    scores += a;
    transforms += (int64_t)pa * 12;
    inliers += pa;
    out_transform += 16 * (int64_t)blockIdx.x;
    if (out_seg_rows && threadIdx.x == 0) {
      out_seg_rows[blockIdx.x] = a;
      if (blockIdx.x == gridDim.x - 1) out_seg_rows[gridDim.x] = a + C;
    }
    if (C == 0) {  // a pair without correspondences: identity (the single-pair entry point refuses this case)
      if (threadIdx.x < 16) out_transform[threadIdx.x] = threadIdx.x % 5 == 0 ? 1.0f : 0.0f;
      return;
    }
  }
  {
    // first maximum, like argmax: key = (inliers + 1, ~index) so that the largest key is the lowest index of the largest
    // count (inliers = -1 marks an invalid hypothesis: key 0 .. never beats a valid one, best stays -1)
    unsigned long long key = 0ull;
    for (int p = threadIdx.x; p < B; p += LG_RT) {
      const int c = inliers[p];
      if (c >= 0) {
        const unsigned long long k = ((unsigned long long)(unsigned)(c + 1) << 32) | (0xffffffffu - (unsigned)p);
        key = k > key ? k : key;
      }
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      const unsigned long long o = __shfl_xor(key, d, WAVE);
      key = o > key ? o : key;
    }
    unsigned long long* kw = reinterpret_cast<unsigned long long*>(sh);
    if ((threadIdx.x & (WAVE - 1)) == 0) kw[threadIdx.x / WAVE] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = 0ull;
      for (int w = 0; w < LG_RT / WAVE; ++w) m = kw[w] > m ? kw[w] : m;
      best_sh = m ? (int)(0xffffffffu - (unsigned)(m & 0xffffffffull)) : -1;
    }
    __syncthreads();
  }
  const int best = best_sh;
  if (best >= 0) {
    if (threadIdx.x < 12) T[threadIdx.x] = transforms[best * 12 + threadIdx.x];
    __syncthreads();
  } else {
    // degenerate: no patch qualifies -> all correspondences, plain scores (:176-180)
    block_procrustes<LG_RT>(src, ref, 0, C, [&](int i) { return (double)fmaxf(scores[i], 0.0f); }, 1e-5f, sh, T);
  }
  for (int s = 0; s < steps; ++s) {
    // scores * inlier mask of the current transform, then weighted Kabsch (:183-190)
    float Tc[12];
    for (int k = 0; k < 12; ++k) Tc[k] = T[k];
    __syncthreads();
    block_procrustes<LG_RT>(src, ref, 0, C,
                     [&](int i) { return inlier(Tc, src, ref, i, radius) ? (double)fmaxf(scores[i], 0.0f) : 0.0; },
                     1e-5f, sh, T);
  }
  if (threadIdx.x < 16) {
    const int r = threadIdx.x / 4, c = threadIdx.x % 4;
    out_transform[threadIdx.x] = r < 3 ? (c < 3 ? T[r * 3 + c] : T[9 + r]) : (c == 3 ? 1.0f : 0.0f);
  }
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_lgr_workspace_bytes(int64_t batch) {
  if (batch < 0) return 0;
  return align_up((size_t)batch * 12 * 4, 256) + 2 * align_up((size_t)batch * 4, 256) + 256;
}

extern "C" int gr_lgr_register_verify(const float* ref_corr_points, const float* src_corr_points,
                                      const float* corr_scores, int64_t num_corr, int64_t batch, const void* pm_ws,
                                      const float* verify_ref_points, const float* verify_src_points,
                                      const float* verify_scores, int64_t num_verify, float acceptance_radius,
                                      int correspondence_threshold, int num_refinement_steps, float* out_transform,
                                      void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(num_corr >= 0 && batch >= 0 && num_refinement_steps >= 1 && num_corr < (1ll << 31), "bad arguments");
  GR_REQUIRE(out_transform != nullptr, "out_transform is null");
  GR_REQUIRE(num_corr > 0, "no correspondences: the reference's procrustes would divide by eps here");
  GR_REQUIRE(ref_corr_points && src_corr_points && corr_scores && pm_ws, "null argument");
  GR_REQUIRE(verify_ref_points && verify_src_points && verify_scores && num_verify > 0 && num_verify <= num_corr,
             "bad verification set");
  if (!ws || ws_bytes < gr_lgr_workspace_bytes(batch)) {
    set_error("lgr workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver c(ws);
  float* transforms = c.take<float>(batch * 12);
  int32_t* valid = c.take<int32_t>(batch);
  int32_t* inl = c.take<int32_t>(batch);
  const int32_t* counts = static_cast<const int32_t*>(pm_ws);  // layout of gr_corr_matrix's workspace
  const int32_t* offsets = counts + batch;
  KernelTimer timer("lgr", stream);
  if (batch > 0) {
    // hypotheses come from ALL correspondences of a patch (:158-164); they are scored on the verification set (:165-170)
    hipLaunchKernelGGL(lgr_local_kernel, dim3((unsigned)batch), dim3(LG_T), 0, stream, src_corr_points, ref_corr_points,
                       corr_scores, counts, offsets, correspondence_threshold, transforms, valid);
    hipLaunchKernelGGL(lgr_verify_kernel, dim3((unsigned)batch), dim3(LG_T), 0, stream, verify_src_points,
                       verify_ref_points, (int)num_verify, transforms, valid, acceptance_radius, inl,
                       (const int32_t*)nullptr, 0, (const int32_t*)nullptr);
  }
  // the first refinement step of the reference is "procrustes with the best hypothesis' mask" (:183), the
  // remaining num_refinement_steps - 1 recompute the mask from the running estimate (:184-190)
  hipLaunchKernelGGL(lgr_refine_kernel, dim3(1), dim3(LG_RT), 0, stream, verify_src_points, verify_ref_points,
                     verify_scores, (int)num_verify, (int)batch, transforms, inl, acceptance_radius, num_refinement_steps,
                     out_transform, (const int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

extern "C" int gr_lgr_register(const float* ref_corr_points, const float* src_corr_points, const float* corr_scores,
                               int64_t num_corr, int64_t batch, const void* pm_ws, float acceptance_radius,
                               int correspondence_threshold, int num_refinement_steps, float* out_transform, void* ws,
                               size_t ws_bytes, void* stream_) {
  return gr_lgr_register_verify(ref_corr_points, src_corr_points, corr_scores, num_corr, batch, pm_ws, ref_corr_points,
                                src_corr_points, corr_scores, num_corr, acceptance_radius, correspondence_threshold,
                                num_refinement_steps, out_transform, ws, ws_bytes, stream_);
}

// Stack mode over scene pairs (test.py:146-212 runs local_global_registration.py:135-193 once per pair): the `batch` patches
// and their correspondences (gr_corr_matrix / gr_corr_gather over ALL patches of the batch) belong to `nseg` pairs, pair s
// owning patches [seg_patch_off[s], seg_patch_off[s + 1]) (device int32, nseg + 1 entries).  out_transforms: nseg x 16
// floats; out_seg_rows (optional, device int32[nseg + 1]): first correspondence row of every pair, last entry = num_corr.
// A pair without correspondences gets the identity.  Three launches for the whole batch, no host synchronisation.
extern "C" int gr_lgr_register_seg(const float* ref_corr_points, const float* src_corr_points, const float* corr_scores,
                                   int64_t num_corr, int64_t batch, const void* pm_ws, const int32_t* seg_patch_off,
                                   int64_t nseg, float acceptance_radius, int correspondence_threshold,
                                   int num_refinement_steps, float* out_transforms, int32_t* out_seg_rows, void* ws,
                                   size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(num_corr >= 0 && batch >= 0 && nseg >= 0 && num_refinement_steps >= 1 && num_corr < (1ll << 31) &&
             nseg < (1 << 24), "bad arguments");
  if (nseg == 0) return GR_OK;
  GR_REQUIRE(out_transforms && pm_ws && seg_patch_off, "null argument");
  GR_REQUIRE(num_corr == 0 || (ref_corr_points && src_corr_points && corr_scores), "null argument");
  if (!ws || ws_bytes < gr_lgr_workspace_bytes(batch)) {
    set_error("lgr workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver c(ws);
  float* transforms = c.take<float>(batch * 12);
  int32_t* valid = c.take<int32_t>(batch);
  int32_t* inl = c.take<int32_t>(batch);
  const int32_t* counts = static_cast<const int32_t*>(pm_ws);  // layout of gr_corr_matrix's workspace:
  const int32_t* offsets = counts + batch;                     // counts[batch], offsets[batch], total -- offsets[batch] = total
  KernelTimer timer("lgr", stream);
  if (batch > 0) {
    hipLaunchKernelGGL(lgr_local_kernel, dim3((unsigned)batch), dim3(LG_T), 0, stream, src_corr_points, ref_corr_points,
                       corr_scores, counts, offsets, correspondence_threshold, transforms, valid);
    // many hypotheses in the call: one thread each over slices of the correspondences (same counts)
    if (batch >= 2048) {
      GR_HIP(hipMemsetAsync(inl, 0, sizeof(int32_t) * batch, stream));
      hipLaunchKernelGGL(lgr_verify_wide_kernel, dim3(LGV_SLICES, (unsigned)nseg), dim3(LG_T), 0, stream, src_corr_points,
                         ref_corr_points, transforms, valid, acceptance_radius, inl, seg_patch_off, offsets);
    } else
    hipLaunchKernelGGL(lgr_verify_kernel, dim3((unsigned)batch), dim3(LG_T), 0, stream, src_corr_points, ref_corr_points,
                       0, transforms, valid, acceptance_radius, inl, seg_patch_off, (int)nseg, offsets);
  }
  hipLaunchKernelGGL(lgr_refine_kernel, dim3((unsigned)nseg), dim3(LG_RT), 0, stream, src_corr_points, ref_corr_points,
                     corr_scores, 0, 0, transforms, inl, acceptance_radius, num_refinement_steps, out_transforms,
                     seg_patch_off, offsets, out_seg_rows);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
