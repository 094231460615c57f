// RANSAC over putative correspondences with a similarity (rotation + uniform scale + translation) model
// ("next" row).  Stands in for the Open3D call GaussReg ends its coarse registration
// with -- geotransformer/utils/open3d.py:169-198 registration_ransac_based_on_correspondence(...,
// TransformationEstimationPointToPoint(True), ransac_n=5, RANSACConvergenceCriteria(10000, 10000)), called
// from experiments/geotransformer.gaussian_splatting.indoor/model.py:209-215.  PARITY UNPINNED: Open3D
// (0.11.2, environment.yaml:111) is neither in the reference tree nor installed, and its sampler is unseeded;
// what is restated is the published algorithm: sample n correspondences, Umeyama similarity, score by the
// number of correspondences within the distance threshold (ties: lower inlier RMSE), keep the best.
//   hypotheses  one thread per hypothesis: counter-hash sampling of n distinct correspondences, Umeyama in
//               fp64 (Horn quaternion rotation, scale = sum(ref_c . R src_c) / sum |src_c|^2)   -> 3x4 transform
//   score       16 lanes per hypothesis share the correspondences (staged through LDS), counts and squared
//               errors are added in a fixed order                                          -> (inliers, rmse)
//               (10 000 hypotheses are 157 waves; scoring inside the hypothesis thread left most of the GPU idle)
//   best        block reduction over hypotheses; optional refit on the best hypothesis' inliers
#include "common.hpp"

namespace gr {
namespace {

constexpr int RS_T = 64;      // hypothesis kernel: a wave per workgroup, so that 10 000 hypotheses spread over 157 CUs
constexpr int RS_MAXN = 8;
constexpr int RS_LPH = 16;    // lanes per hypothesis in the scoring kernel
constexpr int RS_ST = 256;    // scoring kernel threads (16 hypotheses per workgroup)
constexpr int RS_CHUNK = 1024;  // correspondences staged per round
constexpr int64_t RS_WIDE_MIN = 256 * 256 * 2;  // hypotheses per call from which one thread each fills the device

__host__ __device__ inline uint32_t rs_hash(uint32_t seed, uint32_t h, uint32_t k, uint32_t attempt) {
  uint32_t x = seed ^ (h * 0x9E3779B9u) ^ (k * 0x85EBCA6Bu) ^ (attempt * 0xC2B2AE35u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ void jacobi_maxvec4(double A[4][4], double* q) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  double fro2 = 0.0;  // stop at fp64 resolution of the matrix (see lgr.hip)
  for (int p = 0; p < 4; ++p)
    for (int r = 0; r < 4; ++r) fro2 += A[p][r] * A[p][r];
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 4; ++p)
      for (int r = p + 1; r < 4; ++r) off += A[p][r] * A[p][r];
    if (off <= 1e-30 * fro2) break;
    for (int p = 0; p < 3; ++p)
      for (int r = p + 1; r < 4; ++r) {
        if (fabs(A[p][r]) < 1e-300) continue;
        const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) { const double a = A[k][p], b = A[k][r]; A[k][p] = c * a - s * b; A[k][r] = s * a + c * b; }
        for (int k = 0; k < 4; ++k) { const double a = A[p][k], b = A[r][k]; A[p][k] = c * a - s * b; A[r][k] = s * a + c * b; }
        for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][r]; V[k][p] = c * a - s * b; V[k][r] = s * a + c * b; }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (A[k][k] > A[best][best]) best = k;
  double n = 0;
  for (int k = 0; k < 4; ++k) { q[k] = V[k][best]; n += q[k] * q[k]; }
  n = sqrt(n);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}

// Umeyama similarity from sums: n, sum src, sum ref, sum src_a*ref_b (9), sum |src|^2  ->  T (3x4: sR | t)
__device__ bool umeyama_from_sums(double n, const double* ss, const double* sr, const double* sxy, double s2, int with_scale,
                                  float* T) {
  if (n < 1.0) return false;
  double cs[3], cr[3], H[9];
  for (int k = 0; k < 3; ++k) { cs[k] = ss[k] / n; cr[k] = sr[k] / n; }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) H[a * 3 + b] = sxy[a * 3 + b] - n * cs[a] * cr[b];  // sum (src-cs)_a (ref-cr)_b
  const double var = s2 - n * (cs[0] * cs[0] + cs[1] * cs[1] + cs[2] * cs[2]);
  double A[4][4] = {{H[0] + H[4] + H[8], H[5] - H[7], H[6] - H[2], H[1] - H[3]},
                    {H[5] - H[7], H[0] - H[4] - H[8], H[1] + H[3], H[6] + H[2]},
                    {H[6] - H[2], H[1] + H[3], -H[0] + H[4] - H[8], H[5] + H[7]},
                    {H[1] - H[3], H[6] + H[2], H[5] + H[7], -H[0] - H[4] + H[8]}};
  double q[4];
  jacobi_maxvec4(A, q);
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                       2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                       2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  double scale = 1.0;
  if (with_scale) {
    double tr = 0;  // sum_i ref_c . (R src_c) = sum_ab R[b][a] H[a][b]
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) tr += R[b * 3 + a] * H[a * 3 + b];
    if (!(var > 1e-300)) return false;
    scale = tr / var;
    if (!(scale > 0.0) || !isfinite(scale)) return false;
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = (float)(scale * R[r * 3 + c]);
    T[r * 4 + 3] = (float)(cr[r] - scale * (R[r * 3] * cs[0] + R[r * 3 + 1] * cs[1] + R[r * 3 + 2] * cs[2]));
  }
  return true;
}

__global__ __launch_bounds__(RS_T) void ransac_hypotheses_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                                 int C, int n_sample, int num_hyp, uint32_t seed,
                                                                 float thr, int with_scale, float* __restrict__ transforms,
                                                                 int32_t* __restrict__ inliers, float* __restrict__ sqerr,
                                                                 const int32_t* __restrict__ seg_row_off) {
  if (seg_row_off) {  // blockIdx.y = scene pair: its correspondences, its share of the hypothesis tables, its own seed
    const int a = seg_row_off[blockIdx.y];
    C = seg_row_off[blockIdx.y + 1] - a;
    src += 3 * (int64_t)a;
    ref += 3 * (int64_t)a;
    transforms += (int64_t)blockIdx.y * num_hyp * 12;
    inliers += (int64_t)blockIdx.y * num_hyp;
    sqerr += (int64_t)blockIdx.y * num_hyp;
    seed += blockIdx.y;
  }
  const int h = blockIdx.x * RS_T + threadIdx.x;
  float T[12];
  bool ok = false;
  if (h < num_hyp && C >= n_sample) {  // (fewer correspondences than one sample: every hypothesis invalid)
    int idx[RS_MAXN];
    for (int k = 0; k < n_sample; ++k) {
      uint32_t attempt = 0;
      for (;;) {
        idx[k] = (int)(rs_hash(seed, (uint32_t)h, (uint32_t)k, attempt) % (uint32_t)C);
        bool dup = false;
        for (int j = 0; j < k; ++j) dup = dup || idx[j] == idx[k];
        if (!dup || attempt > 64) break;
        ++attempt;
      }
    }
    double ss[3] = {0, 0, 0}, sr[3] = {0, 0, 0}, sxy[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, s2 = 0;
    for (int k = 0; k < n_sample; ++k) {
      const double a[3] = {src[3 * idx[k]], src[3 * idx[k] + 1], src[3 * idx[k] + 2]};
      const double b[3] = {ref[3 * idx[k]], ref[3 * idx[k] + 1], ref[3 * idx[k] + 2]};
      for (int p = 0; p < 3; ++p) {
        ss[p] += a[p];
        sr[p] += b[p];
        s2 += a[p] * a[p];
        for (int r = 0; r < 3; ++r) sxy[p * 3 + r] += a[p] * b[r];
      }
    }
    ok = umeyama_from_sums((double)n_sample, ss, sr, sxy, s2, with_scale, T);
  }
  if (h < num_hyp) {
    inliers[h] = ok ? 0 : -1;
    sqerr[h] = 0.f;
    for (int k = 0; k < 12; ++k) transforms[h * 12 + k] = ok ? T[k] : 0.f;
  }
}

__global__ __launch_bounds__(RS_ST) void ransac_score_kernel(const float* __restrict__ src, const float* __restrict__ ref, int C,
                                                            int num_hyp, float thr, const float* __restrict__ transforms,
                                                            int32_t* __restrict__ inliers, float* __restrict__ sqerr,
                                                            const int32_t* __restrict__ seg_row_off) {
  __shared__ float s_src[RS_CHUNK * 3];
  __shared__ float s_ref[RS_CHUNK * 3];
  if (seg_row_off) {
    const int a = seg_row_off[blockIdx.y];
    C = seg_row_off[blockIdx.y + 1] - a;
    src += 3 * (int64_t)a;
    ref += 3 * (int64_t)a;
    transforms += (int64_t)blockIdx.y * num_hyp * 12;
    inliers += (int64_t)blockIdx.y * num_hyp;
    sqerr += (int64_t)blockIdx.y * num_hyp;
  }
  const int h = blockIdx.x * (RS_ST / RS_LPH) + threadIdx.x / RS_LPH, sub = threadIdx.x % RS_LPH;
  const bool ok = h < num_hyp && inliers[h] >= 0;
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = ok ? transforms[h * 12 + k] : 0.f;
  int cnt = 0;
  float err = 0.f;
  const float thr2 = thr * thr;
  for (int c0 = 0; c0 < C; c0 += RS_CHUNK) {
    const int nn = min(RS_CHUNK, C - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nn * 3; e += RS_ST) {
      s_src[e] = src[3 * (int64_t)c0 + e];
      s_ref[e] = ref[3 * (int64_t)c0 + e];
    }
    __syncthreads();
    if (ok) {
      for (int i = sub; i < nn; i += RS_LPH) {
        const float x = s_src[3 * i], y = s_src[3 * i + 1], z = s_src[3 * i + 2];
        const float dx = s_ref[3 * i] - (fmaf(T[0], x, fmaf(T[1], y, fmaf(T[2], z, T[3]))));
        const float dy = s_ref[3 * i + 1] - (fmaf(T[4], x, fmaf(T[5], y, fmaf(T[6], z, T[7]))));
        const float dz = s_ref[3 * i + 2] - (fmaf(T[8], x, fmaf(T[9], y, fmaf(T[10], z, T[11]))));
        const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (d2 < thr2) {
          ++cnt;
          err += d2;
        }
      }
    }
  }
  // 16 lanes -> one (count, error): pairwise tree in a fixed order
#pragma unroll
  for (int d = RS_LPH / 2; d > 0; d >>= 1) {
    cnt += __shfl_xor(cnt, d, RS_LPH);
    err += __shfl_xor(err, d, RS_LPH);
  }
  if (ok && sub == 0) {
    inliers[h] = cnt;
    sqerr[h] = err;
  }
}

// The same scores, bit for bit, for calls with enough hypotheses to fill the device with one THREAD per hypothesis (the
// stack-mode call: 64 pairs x 10 000): the transform stays in registers, every lane of a wave reads the SAME correspondence --
// LDS broadcasts of coordinate planes, two correspondences per 8-byte read and per packed fp32 instruction -- instead of sixteen
// lanes of a hypothesis walking sixteen different LDS rows (six 4-byte reads per 18 arithmetic instructions: the LDS pipe
// was the bound, 3.6 ms per 64-pair batch).  The order of the error sums is the sixteen-lane kernel's: sixteen partial
// sums over i = r, r + 16, ... in ascending order, then the pairwise tree of its lane butterfly -- ties between hypotheses
// are broken by these sums, and a batch must pick what the single calls pick.
typedef float rs_f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(RS_ST) void ransac_score_wide_kernel(const float* __restrict__ src, const float* __restrict__ ref, int C,
                                                                 int num_hyp, float thr, const float* __restrict__ transforms,
                                                                 int32_t* __restrict__ inliers, float* __restrict__ sqerr,
                                                                 const int32_t* __restrict__ seg_row_off) {
  __shared__ __attribute__((aligned(16))) float s_c[6][RS_CHUNK];  // planes: src x, y, z, ref x, y, z
  static_assert(RS_CHUNK % RS_LPH == 0 && RS_LPH == 16, "a chunk holds whole groups of sixteen: index mod 16 = position mod 16");
  if (seg_row_off) {
    const int a = seg_row_off[blockIdx.y];
    C = seg_row_off[blockIdx.y + 1] - a;
    src += 3 * (int64_t)a;
    ref += 3 * (int64_t)a;
    transforms += (int64_t)blockIdx.y * num_hyp * 12;
    inliers += (int64_t)blockIdx.y * num_hyp;
    sqerr += (int64_t)blockIdx.y * num_hyp;
  }
  const int h = blockIdx.x * RS_ST + threadIdx.x;
  const bool ok = h < num_hyp && inliers[h] >= 0;
  rs_f2 T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float t = ok ? transforms[h * 12 + k] : 0.f;
    T[k] = rs_f2{t, t};
  }
  int cnt = 0;
  rs_f2 acc[RS_LPH / 2];  // acc[m] = the partial sums of positions 2 m and 2 m + 1 (mod 16)
#pragma unroll
  for (int m = 0; m < RS_LPH / 2; ++m) acc[m] = rs_f2{0.f, 0.f};
  const float thr2 = thr * thr;
  for (int c0 = 0; c0 < C; c0 += RS_CHUNK) {
    const int nn = min(RS_CHUNK, C - c0);
    const int np = (nn + RS_LPH - 1) / RS_LPH * RS_LPH;
    __syncthreads();
    for (int e = threadIdx.x; e < nn * 3; e += RS_ST) {
      const int i = e / 3, a = e - 3 * i;
      s_c[a][i] = src[3 * (int64_t)c0 + e];
      s_c[3 + a][i] = ref[3 * (int64_t)c0 + e];
    }
    // the tail of the last group: entries that are inliers of nothing (reference at infinity)
    for (int i = nn + threadIdx.x; i < np; i += RS_ST) {
      s_c[0][i] = s_c[1][i] = s_c[2][i] = 0.f;
      s_c[3][i] = s_c[4][i] = s_c[5][i] = INFINITY;
    }
    __syncthreads();
    if (ok) {
      for (int i0 = 0; i0 < np; i0 += RS_LPH) {
#pragma unroll
        for (int m = 0; m < RS_LPH / 2; ++m) {
          const int i = i0 + 2 * m;
          const rs_f2 x = *reinterpret_cast<const rs_f2*>(&s_c[0][i]), y = *reinterpret_cast<const rs_f2*>(&s_c[1][i]);
          const rs_f2 z = *reinterpret_cast<const rs_f2*>(&s_c[2][i]);
          const rs_f2 rx = *reinterpret_cast<const rs_f2*>(&s_c[3][i]), ry = *reinterpret_cast<const rs_f2*>(&s_c[4][i]);
          const rs_f2 rz = *reinterpret_cast<const rs_f2*>(&s_c[5][i]);
          // (the sixteen-lane kernel's expressions, two correspondences per instruction)
          const rs_f2 dx = rx - __builtin_elementwise_fma(T[0], x, __builtin_elementwise_fma(T[1], y, __builtin_elementwise_fma(T[2], z, T[3])));
          const rs_f2 dy = ry - __builtin_elementwise_fma(T[4], x, __builtin_elementwise_fma(T[5], y, __builtin_elementwise_fma(T[6], z, T[7])));
          const rs_f2 dz = rz - __builtin_elementwise_fma(T[8], x, __builtin_elementwise_fma(T[9], y, __builtin_elementwise_fma(T[10], z, T[11])));
          const rs_f2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
          const bool in0 = d2.x < thr2, in1 = d2.y < thr2;
          cnt += (in0 ? 1 : 0) + (in1 ? 1 : 0);
          acc[m] += rs_f2{in0 ? d2.x : 0.f, in1 ? d2.y : 0.f};  // (+ 0 leaves a sum as it is: same bits as not adding)
        }
      }
    }
  }
  if (ok) {
    // the lane butterfly of the sixteen-lane kernel: partner distance 8, 4, 2, 1
    float a[RS_LPH];
#pragma unroll
    for (int m = 0; m < RS_LPH / 2; ++m) a[2 * m] = acc[m].x, a[2 * m + 1] = acc[m].y;
#pragma unroll
    for (int d = RS_LPH / 2; d > 0; d >>= 1)
#pragma unroll
      for (int j = 0; j < d; ++j) a[j] = a[j] + a[j + d];
    inliers[h] = cnt;
    sqerr[h] = a[0];
  }
}

// best hypothesis (most inliers, then lowest squared error, then lowest index) + optional refit on its inliers
__global__ __launch_bounds__(1024) void ransac_best_kernel(const float* __restrict__ src, const float* __restrict__ ref, int C,
                                                           int num_hyp, const float* __restrict__ transforms,
                                                           const int32_t* __restrict__ inliers,
                                                           const float* __restrict__ sqerr, float thr, int with_scale,
                                                           int refine, float* __restrict__ out /* 4x4 */,
                                                           int32_t* __restrict__ out_stats /* [2]: inliers, best id */,
                                                           const int32_t* __restrict__ seg_row_off, int n_sample,
                                                           const float* __restrict__ fallback) {
  if (seg_row_off) {  // blockIdx.x = scene pair
    const int a = seg_row_off[blockIdx.x];
    C = seg_row_off[blockIdx.x + 1] - a;
    src += 3 * (int64_t)a;
    ref += 3 * (int64_t)a;
    transforms += (int64_t)blockIdx.x * num_hyp * 12;
    inliers += (int64_t)blockIdx.x * num_hyp;
    sqerr += (int64_t)blockIdx.x * num_hyp;
    out += 16 * (int64_t)blockIdx.x;
    if (out_stats) out_stats += 2 * (int64_t)blockIdx.x;
    if (C < n_sample) {  // model.py:209-220 keeps the LocalGlobalRegistration estimate for such a pair
      if (threadIdx.x < 16) out[threadIdx.x] = fallback ? fallback[16 * (int64_t)blockIdx.x + threadIdx.x] : (threadIdx.x % 5 == 0 ? 1.f : 0.f);
      if (threadIdx.x == 0 && out_stats) out_stats[0] = -1, out_stats[1] = -1;
      return;
    }
  }
  __shared__ int s_cnt[1024];
  __shared__ float s_err[1024];
  __shared__ int s_id[1024];
  __shared__ double s_acc[17][1024 / WAVE];
  __shared__ float T[12];
  int bc = -2, bi = -1;
  float be = INFINITY;
  for (int h = threadIdx.x; h < num_hyp; h += 1024) {
    const int c = inliers[h];
    const float e = sqerr[h];
    if (c > bc || (c == bc && e < be)) { bc = c; be = e; bi = h; }
  }
  s_cnt[threadIdx.x] = bc; s_err[threadIdx.x] = be; s_id[threadIdx.x] = bi;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1) {
    if (threadIdx.x < d) {
      const int o = threadIdx.x + d;
      const bool better = s_cnt[o] > s_cnt[threadIdx.x] ||
                          (s_cnt[o] == s_cnt[threadIdx.x] && (s_err[o] < s_err[threadIdx.x] ||
                                                               (s_err[o] == s_err[threadIdx.x] && s_id[o] >= 0 &&
                                                                (s_id[threadIdx.x] < 0 || s_id[o] < s_id[threadIdx.x]))));
      if (better) { s_cnt[threadIdx.x] = s_cnt[o]; s_err[threadIdx.x] = s_err[o]; s_id[threadIdx.x] = s_id[o]; }
    }
    __syncthreads();
  }
  const int best = s_id[0];
  if (threadIdx.x < 12) T[threadIdx.x] = best >= 0 && s_cnt[0] >= 0 ? transforms[best * 12 + threadIdx.x] : (threadIdx.x % 5 == 0 ? 1.f : 0.f);
  __syncthreads();
  if (refine && best >= 0 && s_cnt[0] >= 3) {
    // Umeyama on every inlier of the best hypothesis: 17 fp64 sums {n, sum src, sum ref, sum src_a ref_b, sum |src|^2}
    double acc[17];
    for (int k = 0; k < 17; ++k) acc[k] = 0;
    const float thr2 = thr * thr;
    for (int i = threadIdx.x; i < C; i += 1024) {
      const float x = src[3 * (int64_t)i], y = src[3 * (int64_t)i + 1], z = src[3 * (int64_t)i + 2];
      const float rx = ref[3 * (int64_t)i], ry = ref[3 * (int64_t)i + 1], rz = ref[3 * (int64_t)i + 2];
      const float dx = rx - fmaf(T[0], x, fmaf(T[1], y, fmaf(T[2], z, T[3])));
      const float dy = ry - fmaf(T[4], x, fmaf(T[5], y, fmaf(T[6], z, T[7])));
      const float dz = rz - fmaf(T[8], x, fmaf(T[9], y, fmaf(T[10], z, T[11])));
      if (fmaf(dx, dx, fmaf(dy, dy, dz * dz)) < thr2) {
        const double a3[3] = {x, y, z}, b3[3] = {rx, ry, rz};
        acc[0] += 1.0;
        for (int p = 0; p < 3; ++p) {
          acc[1 + p] += a3[p];
          acc[4 + p] += b3[p];
          acc[16] += a3[p] * a3[p];
          for (int r = 0; r < 3; ++r) acc[7 + p * 3 + r] += a3[p] * b3[r];
        }
      }
    }
    double red[17];
    for (int k = 0; k < 17; ++k) acc[k] = wave_sum_f64_dpp(acc[k]);
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0)
      for (int k = 0; k < 17; ++k) s_acc[k][threadIdx.x / WAVE] = acc[k];
    __syncthreads();
    for (int k = 0; k < 17; ++k) {
      double t = 0;
      for (int w = 0; w < 1024 / WAVE; ++w) t += s_acc[k][w];
      red[k] = t;
    }
    if (threadIdx.x == 0) {
      float Tn[12];
      if (umeyama_from_sums(red[0], red + 1, red + 4, red + 7, red[16], with_scale, Tn))
        for (int k = 0; k < 12; ++k) T[k] = Tn[k];
    }
    __syncthreads();
  }
  if (threadIdx.x < 16) {
    const int r = threadIdx.x / 4, c = threadIdx.x % 4;
    out[threadIdx.x] = r < 3 ? T[r * 4 + c] : (c == 3 ? 1.f : 0.f);
  }
  if (threadIdx.x == 0 && out_stats) {
    out_stats[0] = s_cnt[0];
    out_stats[1] = best;
  }
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" uint32_t gr_ransac_sample_hash(uint32_t seed, uint32_t hypothesis, uint32_t k, uint32_t attempt) {
  return rs_hash(seed, hypothesis, k, attempt);
}

extern "C" size_t gr_ransac_workspace_bytes(int64_t num_hypotheses) {
  if (num_hypotheses < 0) return 0;
  return align_up((size_t)num_hypotheses * 12 * 4, 256) + 2 * align_up((size_t)num_hypotheses * 4, 256) + 256;
}

extern "C" int gr_ransac_similarity(const float* src_points, const float* ref_points, int64_t num_corr, int ransac_n,
                                    int64_t num_hypotheses, uint32_t seed, float distance_threshold, int with_scaling,
                                    int refine, float* out_transform, int32_t* out_stats, void* ws, size_t ws_bytes,
                                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(num_corr >= 0 && num_hypotheses >= 1 && num_hypotheses < (1 << 24), "bad sizes");
  GR_REQUIRE(ransac_n >= 3 && ransac_n <= RS_MAXN, "ransac_n must be in [3, %d]", RS_MAXN);
  GR_REQUIRE(num_corr >= ransac_n, "need at least ransac_n correspondences");
  GR_REQUIRE(src_points && ref_points && out_transform, "null argument");
  if (!ws || ws_bytes < gr_ransac_workspace_bytes(num_hypotheses)) {
    set_error("ransac workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver c(ws);
  float* transforms = c.take<float>(num_hypotheses * 12);
  int32_t* inl = c.take<int32_t>(num_hypotheses);
  float* err = c.take<float>(num_hypotheses);
  KernelTimer timer("ransac", stream);
  hipLaunchKernelGGL(ransac_hypotheses_kernel, dim3((unsigned)((num_hypotheses + RS_T - 1) / RS_T)), dim3(RS_T), 0, stream,
                     src_points, ref_points, (int)num_corr, ransac_n, (int)num_hypotheses, seed, distance_threshold,
                     with_scaling, transforms, inl, err, (const int32_t*)nullptr);
  constexpr int HPB = RS_ST / RS_LPH;
  if (num_hypotheses >= RS_WIDE_MIN)
    hipLaunchKernelGGL(ransac_score_wide_kernel, dim3((unsigned)((num_hypotheses + RS_ST - 1) / RS_ST)), dim3(RS_ST), 0, stream, src_points,
                       ref_points, (int)num_corr, (int)num_hypotheses, distance_threshold, transforms, inl, err,
                       (const int32_t*)nullptr);
  else
  hipLaunchKernelGGL(ransac_score_kernel, dim3((unsigned)((num_hypotheses + HPB - 1) / HPB)), dim3(RS_ST), 0, stream, src_points,
                     ref_points, (int)num_corr, (int)num_hypotheses, distance_threshold, transforms, inl, err,
                     (const int32_t*)nullptr);
  hipLaunchKernelGGL(ransac_best_kernel, dim3(1), dim3(1024), 0, stream, src_points, ref_points, (int)num_corr,
                     (int)num_hypotheses, transforms, inl, err, distance_threshold, with_scaling, refine, out_transform,
                     out_stats, (const int32_t*)nullptr, ransac_n, (const float*)nullptr);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

// Stack mode over `nseg` scene pairs (model.py:209-215 once per pair of test.py:146-212): pair s owns the correspondence rows
// [seg_row_off[s], seg_row_off[s + 1]) (device int32, nseg + 1 entries, e.g. gr_lgr_register_seg's out_seg_rows) and draws its
// hypotheses with seed + s.  A pair with fewer than ransac_n correspondences keeps fallback_transforms[s] (nseg x 16, e.g.
// the LocalGlobalRegistration estimates; null = identity).  out_transforms: nseg x 16, out_stats (optional): nseg x 2.
// Three launches for the whole batch, no host synchronisation.
extern "C" size_t gr_ransac_seg_workspace_bytes(int64_t num_hypotheses, int64_t nseg) {
  if (num_hypotheses < 0 || nseg < 0) return 0;
  return align_up((size_t)nseg * num_hypotheses * 12 * 4, 256) + 2 * align_up((size_t)nseg * num_hypotheses * 4, 256) + 256;
}

extern "C" int gr_ransac_similarity_seg(const float* src_points, const float* ref_points, const int32_t* seg_row_off,
                                        int64_t nseg, int ransac_n, int64_t num_hypotheses, uint32_t seed,
                                        float distance_threshold, int with_scaling, int refine,
                                        const float* fallback_transforms, float* out_transforms, int32_t* out_stats,
                                        void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(nseg >= 0 && nseg < 65536 && num_hypotheses >= 1 && num_hypotheses < (1 << 24), "bad sizes");
  GR_REQUIRE(ransac_n >= 3 && ransac_n <= RS_MAXN, "ransac_n must be in [3, %d]", RS_MAXN);
  if (nseg == 0) return GR_OK;
  GR_REQUIRE(src_points && ref_points && seg_row_off && out_transforms, "null argument");
  if (!ws || ws_bytes < gr_ransac_seg_workspace_bytes(num_hypotheses, nseg)) {
    set_error("ransac workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver c(ws);
  float* transforms = c.take<float>(nseg * num_hypotheses * 12);
  int32_t* inl = c.take<int32_t>(nseg * num_hypotheses);
  float* err = c.take<float>(nseg * num_hypotheses);
  KernelTimer timer("ransac", stream);
  hipLaunchKernelGGL(ransac_hypotheses_kernel, dim3((unsigned)((num_hypotheses + RS_T - 1) / RS_T), (unsigned)nseg), dim3(RS_T),
                     0, stream, src_points, ref_points, 0, ransac_n, (int)num_hypotheses, seed, distance_threshold,
                     with_scaling, transforms, inl, err, seg_row_off);
  constexpr int HPB = RS_ST / RS_LPH;
  // enough hypotheses for one thread each to fill the device: the thread-per-hypothesis kernel (same scores, bit for bit)
  if (nseg * num_hypotheses >= RS_WIDE_MIN)
    hipLaunchKernelGGL(ransac_score_wide_kernel, dim3((unsigned)((num_hypotheses + RS_ST - 1) / RS_ST), (unsigned)nseg), dim3(RS_ST), 0,
                       stream, src_points, ref_points, 0, (int)num_hypotheses, distance_threshold, transforms, inl, err,
                       seg_row_off);
  else
  hipLaunchKernelGGL(ransac_score_kernel, dim3((unsigned)((num_hypotheses + HPB - 1) / HPB), (unsigned)nseg), dim3(RS_ST), 0,
                     stream, src_points, ref_points, 0, (int)num_hypotheses, distance_threshold, transforms, inl, err,
                     seg_row_off);
  hipLaunchKernelGGL(ransac_best_kernel, dim3((unsigned)nseg), dim3(1024), 0, stream, src_points, ref_points, 0,
                     (int)num_hypotheses, transforms, inl, err, distance_threshold, with_scaling, refine, out_transforms,
                     out_stats, seg_row_off, ransac_n, fallback_transforms);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
