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
// Passes: xyz' of cloud 2 in fp64 (12 of the 248 bytes of a vertex) -> centres -> flags -> scan -> one
// transform+gather pass per cloud that reads only the kept vertices and writes them at their final place (the
// transformed cloud 2 is never materialised: a dropped vertex costs 12 bytes of reads, not 496 of traffic).
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
  double T1[9], T2[25], T3[49];  // SH band transforms: new[j] = sum_i old[i] * T[i][j] (fp32 values, widened once)
};

// xyz' of one vertex of cloud 2 in fp64 (like the reference's float32 @ float64 product): ONE expression, used by the
// pass that feeds the centre and by the pass that writes the records, so both see the same bits
__device__ __forceinline__ void transform_xyz(const FuseParams& p, float fx, float fy, float fz, double nx[3]) {
  const double x = fx, y = fy, z = fz;
#pragma unroll
  for (int r = 0; r < 3; ++r) nx[r] = ((x * p.R[r * 3] + y * p.R[r * 3 + 1]) + z * p.R[r * 3 + 2]) * p.scale + p.t[r];
}

__global__ __launch_bounds__(256) void xyz_kernel(const float* __restrict__ in, int n, FuseParams p,
                                                  double* __restrict__ xyz64) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* v = in + (int64_t)i * REC;
  double nx[3];
  transform_xyz(p, v[0], v[1], v[2], nx);
#pragma unroll
  for (int r = 0; r < 3; ++r) xyz64[(int64_t)i * 3 + r] = nx[r];
}

// The record of one vertex of cloud 2, transformed in place (v points at its 62 floats in LDS).
__device__ __forceinline__ void transform_record(const FuseParams& p, float* v) {
  double nx[3];
  transform_xyz(p, v[0], v[1], v[2], nx);
#pragma unroll
  for (int r = 0; r < 3; ++r) v[r] = (float)nx[r];
  // f_dc and the opacity are unchanged (the normals are zeroed when the records are written)
  if (p.scale != 1.0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) v[OFF_SCALE + r] = (float)((double)v[OFF_SCALE + r] + p.log_scale);
  }
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
  for (int k = 0; k < 4; ++k) v[OFF_ROT + k] = cand[k] / den;
  // SH bands (fp64 accumulate like the reference's float64 matmul), per channel; a band is read before it is overwritten
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float* s = v + OFF_REST + c * 15;
    double b1[3], b2[5], b3[7];
#pragma unroll
    for (int io = 0; io < 3; ++io) b1[io] = (double)s[io];
#pragma unroll
    for (int io = 0; io < 5; ++io) b2[io] = (double)s[3 + io];
#pragma unroll
    for (int io = 0; io < 7; ++io) b3[io] = (double)s[8 + io];
#pragma unroll
    for (int jn = 0; jn < 3; ++jn) {
      double a = 0.0;
#pragma unroll
      for (int io = 0; io < 3; ++io) a += b1[io] * p.T1[io * 3 + jn];
      s[jn] = (float)a;
    }
#pragma unroll
    for (int jn = 0; jn < 5; ++jn) {
      double a = 0.0;
#pragma unroll
      for (int io = 0; io < 5; ++io) a += b2[io] * p.T2[io * 5 + jn];
      s[3 + jn] = (float)a;
    }
#pragma unroll
    for (int jn = 0; jn < 7; ++jn) {
      double a = 0.0;
#pragma unroll
      for (int io = 0; io < 7; ++io) a += b3[io] * p.T3[io * 7 + jn];
      s[8 + jn] = (float)a;
    }
  }
}

// Transform (cloud 2 only) + order-preserving gather of whole records.  One wave owns 64 consecutive vertices: their
// 15 872 bytes are ONE contiguous piece of the input, fetched as 16-byte vectors (only the vectors that touch a kept
// vertex) into LDS; kept vertices of cloud 2 are transformed there, one lane each; the kept records of the chunk are
// consecutive in the output (prefix = exclusive scan of the chunk counts, both clouds in one sequence), so they leave as one contiguous run of 8-byte
// vectors.  The normals (columns 3..5) are written as zeros: the reference's save_ply does that whatever the inputs
// held (gs_fusion.py:186-187).
constexpr int FR_WAVES = 4, FR_CHUNK = 64, FR_VEC = FR_CHUNK * REC / 4;  // 992 float4 per chunk
template <bool XF>
__global__ __launch_bounds__(FR_WAVES* WAVE) void fuse_records_kernel(const float* __restrict__ rec, int n,
                                                                      const unsigned long long* __restrict__ keep,
                                                                      const int32_t* __restrict__ prefix, FuseParams p,
                                                                      float* __restrict__ out) {
  __shared__ float4 s_rec[FR_WAVES][FR_VEC];
  __shared__ int s_lane[FR_WAVES][FR_CHUNK];
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int base = (blockIdx.x * FR_WAVES + wave) * FR_CHUNK;
  if (base >= n) return;  // wave-level synchronisation only below
  const unsigned long long mask = keep[base / FR_CHUNK];  // bits of vertices past the end are clear
  if (mask == 0ull) return;
  const int cnt = __popcll(mask);
  const int o0 = prefix[base / FR_CHUNK];
  const bool f = (mask >> lane) & 1ull;
  if (f) s_lane[wave][__popcll(mask & ((1ull << lane) - 1ull))] = lane;
  const int nfl = min(FR_CHUNK, n - base) * REC;  // floats of this chunk that exist
  const float* src = rec + (int64_t)base * REC;
  float* rows = reinterpret_cast<float*>(&s_rec[wave][0]);
  float4 v[(FR_VEC + WAVE - 1) / WAVE];
#pragma unroll
  for (int k = 0; k < (FR_VEC + WAVE - 1) / WAVE; ++k) {
    const int q = k * WAVE + lane;                 // bytes [16 q, 16 q + 16) of the chunk: vertices (2q)/31 .. (2q+1)/31
    const int rlo = min((2 * q) / 31, FR_CHUNK - 1), rhi = min((2 * q + 1) / 31, FR_CHUNK - 1);
    const bool need = q < FR_VEC && (((mask >> rlo) | (mask >> rhi)) & 1ull);
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (need) {
      if (4 * q + 3 < nfl) v[k] = reinterpret_cast<const float4*>(src)[q];
      else {  // the last vector of the last chunk of an odd-sized cloud
        if (4 * q < nfl) v[k].x = src[4 * q];
        if (4 * q + 1 < nfl) v[k].y = src[4 * q + 1];
        if (4 * q + 2 < nfl) v[k].z = src[4 * q + 2];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < (FR_VEC + WAVE - 1) / WAVE; ++k) {
    const int q = k * WAVE + lane;
    if (q < FR_VEC) s_rec[wave][q] = v[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (XF) {
    if (f) transform_record(p, rows + lane * REC);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  float2* dst = reinterpret_cast<float2*>(out + (int64_t)o0 * REC);
  const int ne = cnt * (REC / 2);
#pragma unroll 4
  for (int e = lane; e < ne; e += WAVE) {
    const int s = e / (REC / 2), c2 = e - s * (REC / 2);
    float2 val = *reinterpret_cast<const float2*>(rows + s_lane[wave][s] * REC + 2 * c2);
    if (c2 == 1) val.y = 0.0f;
    if (c2 == 2) val.x = 0.0f, val.y = 0.0f;
    dst[e] = val;
  }
}

// fp64 column sums of an (n,3) array with stride `stride` elements: per-block partials, fixed order
template <typename T>
__global__ __launch_bounds__(256) void centre_partial_kernel(const T* __restrict__ a, int n, int stride,
                                                             double* __restrict__ partial, T* __restrict__ compact) {
  __shared__ double sh[3][256];
  double s[3] = {0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    for (int k = 0; k < 3; ++k) {
      const T v = a[(int64_t)i * stride + k];
      s[k] += (double)v;
      if (compact) compact[(int64_t)i * 3 + k] = v;  // 12 of the 248 bytes of a vertex, so that the selection does not pull the rows in again
    }
  for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d)
      for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void centre_final_kernel(const double* __restrict__ partial, int blocks, int n,
                                                           double* __restrict__ centre) {
  __shared__ double sh[3][256];
  double s[3] = {0, 0, 0};
  for (int b = threadIdx.x; b < blocks; b += 256)
    for (int k = 0; k < 3; ++k) s[k] += partial[b * 3 + k];
  for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {  // fixed order: the same bits every run
    if (threadIdx.x < d)
      for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x < 3) centre[threadIdx.x] = sh[threadIdx.x][0] / (double)n;
}

// keep[i] = own-centre distance < other-centre distance (gs_fusion.py:252-253), fp64.  Written per chunk of 64 vertices
// (one wave): the 64-bit keep mask and its population count -- the scan below runs over chunks, not vertices.
template <typename T>
__global__ __launch_bounds__(256) void select_kernel(const T* __restrict__ xyz, int n, const double* __restrict__ own,
                                                     const double* __restrict__ other,
                                                     unsigned long long* __restrict__ mask, int32_t* __restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool keep = false;
  if (i < n) {
    double d0 = 0, d1 = 0;
    for (int k = 0; k < 3; ++k) {
      const double v = (double)xyz[(int64_t)i * 3 + k];
      d0 += (v - own[k]) * (v - own[k]);
      d1 += (v - other[k]) * (v - other[k]);
    }
    keep = sqrt(d0) < sqrt(d1);
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & (WAVE - 1)) == 0 && i < n) {
    mask[i / WAVE] = m;
    count[i / WAVE] = __popcll(m);
  }
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_gs_fuse_workspace_bytes(int64_t n1, int64_t n2) {
  if (n1 < 0 || n2 < 0) return 0;
  const size_t chunks = (size_t)((n1 + WAVE - 1) / WAVE + (n2 + WAVE - 1) / WAVE);
  return align_up((size_t)n2 * 3 * 8, 256) + align_up((size_t)n1 * 3 * 4, 256) + align_up(chunks * 8, 256) +
         2 * align_up(chunks * 4, 256) + align_up(scan_ws_ints(chunks > 0 ? chunks : 1) * 4, 256) +
         align_up(1024 * 3 * 8 * 2, 256) + 4096;
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
  GR_REQUIRE(((uintptr_t)rec1 & 15) == 0 && ((uintptr_t)rec2 & 15) == 0 && ((uintptr_t)out_rec & 7) == 0,
             "record arrays must be 16-byte aligned (output: 8)");
  if (!ws || ws_bytes < gr_gs_fuse_workspace_bytes(n1, n2)) {
    set_error("gs_fuse workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver c(ws);
  const int64_t nc1 = (n1 + WAVE - 1) / WAVE, nc2 = (n2 + WAVE - 1) / WAVE;
  double* xyz64 = c.take<double>(n2 * 3);
  float* xyz32 = c.take<float>(n1 * 3);
  unsigned long long* keep = c.take<unsigned long long>(nc1 + nc2);
  int32_t* count = c.take<int32_t>(nc1 + nc2);
  int32_t* prefix = c.take<int32_t>(nc1 + nc2);
  int32_t* scan_ws = c.take<int32_t>(scan_ws_ints(nc1 + nc2));
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
  hipLaunchKernelGGL(xyz_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, rec2, (int)n2, p, xyz64);
  const int b1 = (int)std::min<int64_t>(1024, (n1 + 255) / 256), b2 = (int)std::min<int64_t>(1024, (n2 + 255) / 256);
  hipLaunchKernelGGL((centre_partial_kernel<float>), dim3(b1), dim3(256), 0, stream, rec1, (int)n1, REC, partial, xyz32);
  hipLaunchKernelGGL(centre_final_kernel, dim3(1), dim3(256), 0, stream, partial, b1, (int)n1, centres);
  hipLaunchKernelGGL((centre_partial_kernel<double>), dim3(b2), dim3(256), 0, stream, xyz64, (int)n2, 3, partial + 1024 * 3,
                     (double*)nullptr);
  hipLaunchKernelGGL(centre_final_kernel, dim3(1), dim3(256), 0, stream, partial + 1024 * 3, b2, (int)n2, centres + 4);
  hipLaunchKernelGGL((select_kernel<float>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, xyz32, (int)n1, centres,
                     centres + 4, keep, count);
  hipLaunchKernelGGL((select_kernel<double>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, xyz64, (int)n2,
                     centres + 4, centres, keep + nc1, count + nc1);
  GR_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(count, prefix, nc1 + nc2, 1, nc1 + nc2, scan_ws, total, stream);  // over chunks, not vertices
  if (rc != GR_OK) return rc;
  const int per_block = FR_WAVES * FR_CHUNK;
  hipLaunchKernelGGL((fuse_records_kernel<false>), dim3((unsigned)((n1 + per_block - 1) / per_block)), dim3(FR_WAVES * WAVE), 0,
                     stream, rec1, (int)n1, keep, prefix, p, out_rec);
  hipLaunchKernelGGL((fuse_records_kernel<true>), dim3((unsigned)((n2 + per_block - 1) / per_block)), dim3(FR_WAVES * WAVE), 0,
                     stream, rec2, (int)n2, keep + nc1, prefix + nc1, p, out_rec);
  GR_LAUNCH_CHECK();
  int32_t t = 0;
  GR_HIP(hipMemcpyAsync(&t, total, sizeof(t), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  *h_num_out = t;
  return GR_OK;
}
