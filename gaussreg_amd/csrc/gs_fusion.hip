// GS .ply fusion ("next" row): the arithmetic of gs_fusion.py:231-262 on the wire
// format itself -- a vertex is 62 little-endian fp32 {x y z nx ny nz f_dc[3] f_rest[45] opacity scale[3]
// rot[4]} (gs_fusion.py:172-184), f_rest channel-major (P,3,15) (gs_fusion.py:203-215).
//   transform   cloud 2: xyz' = (R xyz) s + t in fp64 (gs_fusion.py:241), log-scales + ln s (:242),
//               rot' = matrix_to_quaternion(R . quaternion_to_matrix(rot)) in fp32 (:243-244, :70-170),
//               SH bands 1..3 rotated by three fixed matrices (3x3, 5x5, 7x7) -- the reference fits them
//               per call from random directions with pinv (:53-68); they only depend on R, so the host
//               computes them once and passes them in
//   centres     fp64 means of cloud 1 and of the transformed cloud 2 (:250-251)
//   select      keep the vertices closer to their own cloud's centre (:252-253), order-preserving
//               compaction (scan) and a gather of whole 248-byte records (:254-260)
#include "common.hpp"

namespace gr {
namespace {

constexpr int REC = 62;  // floats per vertex
constexpr int OFF_REST = 9, OFF_SCALE = 55, OFF_ROT = 58;

struct FuseParams {
  double R[9];     // rotation with the similarity scale divided out (gs_fusion.py:239-240)
  double t[3];
  double scale, log_scale;
  float R32[9];
  float T1[9], T2[25], T3[49];  // SH band transforms: new[j] = sum_i old[i] * T[i][j]
};

__global__ __launch_bounds__(256) void transform_kernel(const float* __restrict__ in, int n, FuseParams p,
                                                        float* __restrict__ out, double* __restrict__ xyz64) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* v = in + (int64_t)i * REC;
  float* o = out + (int64_t)i * REC;
  for (int k = 3; k < 9; ++k) o[k] = v[k];       // f_dc unchanged (the normals are zeroed when the records are gathered)
  o[54] = v[54];                                   // opacity
  // xyz (fp64, like the reference's float32 @ float64 product)
  const double x = v[0], y = v[1], z = v[2];
  double nx[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) nx[r] = ((x * p.R[r * 3] + y * p.R[r * 3 + 1]) + z * p.R[r * 3 + 2]) * p.scale + p.t[r];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    o[r] = (float)nx[r];
    xyz64[(int64_t)i * 3 + r] = nx[r];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) o[OFF_SCALE + r] = p.scale != 1.0 ? (float)((double)v[OFF_SCALE + r] + p.log_scale) : v[OFF_SCALE + r];
  // quaternion (real part first): gs_fusion.py:70-99 quaternion_to_matrix, fp32
  const float qr = v[OFF_ROT], qi = v[OFF_ROT + 1], qj = v[OFF_ROT + 2], qk = v[OFF_ROT + 3];
  const float two_s = 2.0f / (((qr * qr + qi * qi) + qj * qj) + qk * qk);
  float m[9] = {1 - two_s * (qj * qj + qk * qk), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
                two_s * (qi * qj + qk * qr), 1 - two_s * (qi * qi + qk * qk), two_s * (qj * qk - qi * qr),
                two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi * qi + qj * qj)};
  float M[9];  // R32 . m
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[r * 3 + c] = (p.R32[r * 3] * m[c] + p.R32[r * 3 + 1] * m[3 + c]) + p.R32[r * 3 + 2] * m[6 + c];
  // gs_fusion.py:112-170 matrix_to_quaternion
  const float q0 = 1.0f + M[0] + M[4] + M[8], q1 = 1.0f + M[0] - M[4] - M[8];
  const float q2 = 1.0f - M[0] + M[4] - M[8], q3 = 1.0f - M[0] - M[4] + M[8];
  float qa[4] = {q0 > 0 ? sqrtf(q0) : 0.f, q1 > 0 ? sqrtf(q1) : 0.f, q2 > 0 ? sqrtf(q2) : 0.f, q3 > 0 ? sqrtf(q3) : 0.f};
  int best = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (qa[k] > qa[best]) best = k;  // argmax, first maximum
  float cand[4];
  if (best == 0) { cand[0] = qa[0] * qa[0]; cand[1] = M[7] - M[5]; cand[2] = M[2] - M[6]; cand[3] = M[3] - M[1]; }
  else if (best == 1) { cand[0] = M[7] - M[5]; cand[1] = qa[1] * qa[1]; cand[2] = M[3] + M[1]; cand[3] = M[2] + M[6]; }
  else if (best == 2) { cand[0] = M[2] - M[6]; cand[1] = M[3] + M[1]; cand[2] = qa[2] * qa[2]; cand[3] = M[5] + M[7]; }
  else { cand[0] = M[3] - M[1]; cand[1] = M[6] + M[2]; cand[2] = M[7] + M[5]; cand[3] = qa[3] * qa[3]; }
  const float den = 2.0f * fmaxf(qa[best], 0.1f);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[OFF_ROT + k] = cand[k] / den;
  // SH bands (fp64 accumulate like the reference's float64 matmul), per channel
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* s = v + OFF_REST + c * 15;
    float* d = o + OFF_REST + c * 15;
    for (int jn = 0; jn < 3; ++jn) {
      double a = 0.0;
      for (int io = 0; io < 3; ++io) a += (double)s[io] * (double)p.T1[io * 3 + jn];
      d[jn] = (float)a;
    }
    for (int jn = 0; jn < 5; ++jn) {
      double a = 0.0;
      for (int io = 0; io < 5; ++io) a += (double)s[3 + io] * (double)p.T2[io * 5 + jn];
      d[3 + jn] = (float)a;
    }
    for (int jn = 0; jn < 7; ++jn) {
      double a = 0.0;
      for (int io = 0; io < 7; ++io) a += (double)s[8 + io] * (double)p.T3[io * 7 + jn];
      d[8 + jn] = (float)a;
    }
  }
}

// fp64 column sums of an (n,3) array with stride `stride` elements: per-block partials, fixed order
template <typename T>
__global__ __launch_bounds__(256) void centre_partial_kernel(const T* __restrict__ a, int n, int stride,
                                                             double* __restrict__ partial) {
  __shared__ double sh[3][256];
  double s[3] = {0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    for (int k = 0; k < 3; ++k) s[k] += (double)a[(int64_t)i * stride + k];
  for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d)
      for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void centre_final_kernel(const double* __restrict__ partial, int blocks, int n, double* __restrict__ centre) {
  if (threadIdx.x < 3) {
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += partial[b * 3 + threadIdx.x];
    centre[threadIdx.x] = s / (double)n;
  }
}

// flag[i] = own-centre distance < other-centre distance (gs_fusion.py:252-253), fp64
template <typename T>
__global__ __launch_bounds__(256) void select_kernel(const T* __restrict__ xyz, int n, int stride,
                                                     const double* __restrict__ own, const double* __restrict__ other,
                                                     int32_t* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double d0 = 0, d1 = 0;
  for (int k = 0; k < 3; ++k) {
    const double v = (double)xyz[(int64_t)i * stride + k];
    d0 += (v - own[k]) * (v - own[k]);
    d1 += (v - other[k]) * (v - other[k]);
  }
  flag[i] = sqrt(d0) < sqrt(d1) ? 1 : 0;
}

__global__ __launch_bounds__(256) void gather_records_kernel(const float* __restrict__ rec, int n,
                                                             const int32_t* __restrict__ flag,
                                                             const int32_t* __restrict__ offs, int base,
                                                             float* __restrict__ out) {
  // one wave per vertex: 62 floats copied by 62 lanes
  const int i = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  if (i >= n || !flag[i] || lane >= REC) return;
  // columns 3..5 are the normals: the reference's save_ply writes zeros there whatever the inputs held (gs_fusion.py:186-187)
  out[(int64_t)(base + offs[i]) * REC + lane] = (lane >= 3 && lane < 6) ? 0.0f : rec[(int64_t)i * REC + lane];
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_gs_fuse_workspace_bytes(int64_t n1, int64_t n2) {
  if (n1 < 0 || n2 < 0) return 0;
  const int64_t n = n1 + n2;
  return align_up((size_t)n2 * REC * 4, 256) + align_up((size_t)n2 * 3 * 8, 256) + 2 * align_up((size_t)n * 4, 256) +
         align_up(scan_ws_ints(n > 0 ? n : 1) * 4, 256) + align_up(1024 * 3 * 8 * 2, 256) + 4096;
}

extern "C" int gr_gs_fuse(const float* rec1, int64_t n1, const float* rec2, int64_t n2, const double* h_rotation,
                          const double* h_translation, double h_scale, const float* h_sh_t1, const float* h_sh_t2,
                          const float* h_sh_t3, float* out_rec, int64_t* h_num_out, void* ws, size_t ws_bytes,
                          void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_num_out != nullptr, "h_num_out is null");
  *h_num_out = 0;
  GR_REQUIRE(n1 >= 0 && n2 >= 0 && n1 + n2 < (1ll << 31) - 1, "bad sizes");
  GR_REQUIRE(h_rotation && h_translation && h_sh_t1 && h_sh_t2 && h_sh_t3 && h_scale > 0.0, "bad transform arguments");
  if (n1 + n2 == 0) return GR_OK;
  GR_REQUIRE(n1 > 0 && n2 > 0, "both clouds must be non-empty (the reference takes the mean of each)");
  GR_REQUIRE(rec1 && rec2 && out_rec, "null argument");
  if (!ws || ws_bytes < gr_gs_fuse_workspace_bytes(n1, n2)) {
    set_error("gs_fuse workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver c(ws);
  float* rec2t = c.take<float>(n2 * REC);
  double* xyz64 = c.take<double>(n2 * 3);
  int32_t* flag = c.take<int32_t>(n1 + n2);
  int32_t* offs = c.take<int32_t>(n1 + n2);
  int32_t* scan_ws = c.take<int32_t>(scan_ws_ints(n1 + n2));
  double* partial = c.take<double>(1024 * 3 * 2);
  double* centres = c.take<double>(8);
  int32_t* total = c.take<int32_t>(2);
  FuseParams p;
  for (int i = 0; i < 9; ++i) {
    p.R[i] = h_rotation[i];
    p.R32[i] = (float)h_rotation[i];
    p.T1[i] = h_sh_t1[i];
  }
  for (int i = 0; i < 3; ++i) p.t[i] = h_translation[i];
  for (int i = 0; i < 25; ++i) p.T2[i] = h_sh_t2[i];
  for (int i = 0; i < 49; ++i) p.T3[i] = h_sh_t3[i];
  p.scale = h_scale;
  p.log_scale = log(h_scale);
  KernelTimer timer("gs_fuse", stream);
  hipLaunchKernelGGL(transform_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, rec2, (int)n2, p, rec2t,
                     xyz64);
  const int b1 = (int)std::min<int64_t>(1024, (n1 + 255) / 256), b2 = (int)std::min<int64_t>(1024, (n2 + 255) / 256);
  hipLaunchKernelGGL((centre_partial_kernel<float>), dim3(b1), dim3(256), 0, stream, rec1, (int)n1, REC, partial);
  hipLaunchKernelGGL(centre_final_kernel, dim3(1), dim3(64), 0, stream, partial, b1, (int)n1, centres);
  hipLaunchKernelGGL((centre_partial_kernel<double>), dim3(b2), dim3(256), 0, stream, xyz64, (int)n2, 3, partial + 1024 * 3);
  hipLaunchKernelGGL(centre_final_kernel, dim3(1), dim3(64), 0, stream, partial + 1024 * 3, b2, (int)n2, centres + 4);
  hipLaunchKernelGGL((select_kernel<float>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, rec1, (int)n1, REC,
                     centres, centres + 4, flag);
  hipLaunchKernelGGL((select_kernel<double>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, xyz64, (int)n2, 3,
                     centres + 4, centres, flag + n1);
  GR_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(flag, offs, n1 + n2, 1, n1 + n2, scan_ws, total, stream);
  if (rc != GR_OK) return rc;
  hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)((n1 + 3) / 4)), dim3(256), 0, stream, rec1, (int)n1, flag, offs,
                     0, out_rec);
  hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, stream, rec2t, (int)n2, flag + n1,
                     offs + n1, 0, out_rec);
  GR_LAUNCH_CHECK();
  int32_t t = 0;
  GR_HIP(hipMemcpyAsync(&t, total, sizeof(t), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  *h_num_out = t;
  return GR_OK;
}
