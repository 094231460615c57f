// LearnableLogOptimalTransport forward (SuperGlue-style log-domain Sinkhorn with a dustbin), one
// workgroup per score matrix, the whole padded matrix resident in LDS for all iterations.
//
// Replaces  geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66  (forward), which runs
// 2 x num_iterations logsumexp kernels over (B, M+1, N+1) plus ~20 small ATen ops; at the demo
// shapes (B=256, M=N=128, 100 iterations -- config.py:118-125) that is >200 launches re-reading a
// 17 MB tensor each, here one launch that reads the scores once.
#include <algorithm>

#include "common.hpp"

namespace gr {
namespace {

constexpr int SK_PARTS = 4;   // lanes per row / column (adjacent lanes: their partials meet through two DPP steps)
constexpr int SK_T = 512;     // 8 waves, two per SIMD: 128 rows (columns) x 4 lanes; the demo shape's 129th row and column
                              // (the dustbins) are reduced by whole waves on the side
constexpr int SK_MAIN = SK_T / SK_PARTS;  // rows / columns that get their own four lanes
constexpr int SK_MAXD = 144;  // largest M + 1 / N + 1 supported
constexpr int SK_PER = SK_MAXD / SK_PARTS;  // elements of a row / column one lane reduces (<= 36)

__device__ __forceinline__ float lse_finish(float mx, float s) { return logf(s) + mx; }

// row stride of the padded matrix in LDS: the smallest multiple of 4 >= C that is 4 (mod 32)
__host__ __device__ inline int sinkhorn_ld(int C) {
  const int c4 = (C + 3) / 4 * 4;
  return c4 % 32 == 4 ? c4 : c4 + (36 - c4 % 32) % 32;
}

// (max, sum exp(x - max)) of the four adjacent lanes that share a row / column -> logsumexp, valid in all four
__device__ __forceinline__ float quad_xor1(float x) {  // DPP quad_perm:[1,0,3,2]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_xor2(float x) {  // DPP quad_perm:[2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));
}
// wave-wide max / sum of a float on the DPP shift network (see common.hpp), valid in every lane
__device__ __forceinline__ float wave_max_f32_dpp(float x) {
  const int ninf = __float_as_int(-INFINITY);
#define GR_FMAX_STEP(CTRL, ROWMASK) x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(x), CTRL, ROWMASK, 0xf, false)));
  GR_FMAX_STEP(0x111, 0xf) GR_FMAX_STEP(0x112, 0xf) GR_FMAX_STEP(0x114, 0xf) GR_FMAX_STEP(0x118, 0xf)
  GR_FMAX_STEP(0x142, 0xa) GR_FMAX_STEP(0x143, 0xc)
#undef GR_FMAX_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_sum_f32_dpp(float x) {
#define GR_FADD_STEP(CTRL, ROWMASK) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xf, false));
  GR_FADD_STEP(0x111, 0xf) GR_FADD_STEP(0x112, 0xf) GR_FADD_STEP(0x114, 0xf) GR_FADD_STEP(0x118, 0xf)
  GR_FADD_STEP(0x142, 0xa) GR_FADD_STEP(0x143, 0xc)
#undef GR_FADD_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// logsumexp of `len` values base[k * stride] + add[k] by one whole wave (the rows / columns beyond SK_MAIN)
__device__ __forceinline__ float wave_lse(const float* base, int stride, const float* add, int len, int lane) {
  float x[(SK_MAXD + WAVE - 1) / WAVE];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < (SK_MAXD + WAVE - 1) / WAVE; ++t) {
    const int k = lane + t * WAVE;
    const float sv = base[(size_t)k * stride] + add[k];
    x[t] = k < len ? sv : -INFINITY;
    mx = fmaxf(mx, x[t]);
  }
  mx = wave_max_f32_dpp(mx);
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < (SK_MAXD + WAVE - 1) / WAVE; ++t) s += lane + t * WAVE < len ? __expf(x[t] - mx) : 0.f;
  return lse_finish(mx, wave_sum_f32_dpp(s));
}

__device__ __forceinline__ float lse_merge4(float mx, float s) {
  float m = fmaxf(mx, quad_xor1(mx));
  m = fmaxf(m, quad_xor2(m));
  float t = mx > -INFINITY ? s * __expf(mx - m) : 0.f;  // empty part: (-inf, 0)
  t += quad_xor1(t);
  t += quad_xor2(t);
  return lse_finish(m, t);
}

// Lane (idx, part) = (tid / 4, tid % 4).  Row pass: the lane reduces the elements j = part + 4 t of row idx; column pass:
// the rows [36 part, 36 part + 36) of column idx.  With a row stride of 132 floats both patterns touch every LDS bank at
// most twice per wave (the 64-lane minimum): bank = 4 idx + part + 4 t, and 16 part + idx + 4 t (36 * 132 = 16 mod 32).
// The slice stays in registers between the max pass and the sum pass; the four partials meet through DPP, so an iteration
// has two barriers (u complete, v complete) and no partial arrays in LDS.
__global__ __launch_bounds__(SK_T) void sinkhorn_kernel(const float* __restrict__ scores, int M, int N,
                                                        const uint8_t* __restrict__ row_masks,
                                                        const uint8_t* __restrict__ col_masks,
                                                        const float* __restrict__ alpha_p, int iters, float inf,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = M + 1, C = N + 1;
  const int ld = sinkhorn_ld(C);
  float* S = reinterpret_cast<float*>(smem);
  float* u = S + (size_t)R * ld;
  float* v = u + R;
  float* log_mu = v + C;
  float* log_nu = log_mu + R;
  int* cnt = reinterpret_cast<int*>(log_nu + C);
  const int b = blockIdx.x;
  const float alpha = alpha_p[0];
  const uint8_t* rm = row_masks ? row_masks + (int64_t)b * M : nullptr;
  const uint8_t* cm = col_masks ? col_masks + (int64_t)b * N : nullptr;
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  // valid row / column counts (learnable_sinkhorn.py:50-51)
  int nr = 0, nc = 0;
  for (int i = threadIdx.x; i < M; i += SK_T) nr += (!rm || rm[i]) ? 1 : 0;
  for (int j = threadIdx.x; j < N; j += SK_T) nc += (!cm || cm[j]) ? 1 : 0;
  if (nr) atomicAdd(&cnt[0], nr);
  if (nc) atomicAdd(&cnt[1], nc);
  // padded scores: [scores | alpha ; alpha ... alpha], masked rows / columns -> -inf (:44-48)
  for (int e = threadIdx.x; e < R * C; e += SK_T) {
    const int i = e / C, j = e % C;
    float x = (i < M && j < N) ? scores[((int64_t)b * M + i) * N + j] : alpha;
    const bool masked = (i < M && rm && !rm[i]) || (j < N && cm && !cm[j]);
    S[i * ld + j] = masked ? -inf : x;
  }
  __syncthreads();
  const float nvr = (float)cnt[0], nvc = (float)cnt[1];
  const float norm = -logf(nvr + nvc);  // :52
  for (int i = threadIdx.x; i < R; i += SK_T) {
    float x = i < M ? norm : logf(nvc) + norm;         // :54-56
    if (i < M && rm && !rm[i]) x = -inf;               // :57
    log_mu[i] = x;
    u[i] = 0.f;
  }
  for (int j = threadIdx.x; j < C; j += SK_T) {
    float x = j < N ? norm : logf(nvr) + norm;         // :59-61
    if (j < N && cm && !cm[j]) x = -inf;               // :62
    log_nu[j] = x;
    v[j] = 0.f;
  }
  __syncthreads();
  // log_sinkhorn_normalization (:13-18); logsumexp = max + log(sum exp(x - max))
  const int idx = threadIdx.x / SK_PARTS, part = threadIdx.x % SK_PARTS;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int row_len = idx < R ? (C - part + SK_PARTS - 1) / SK_PARTS : 0;       // elements part, part + 4, ... of a row
  const int i0 = part * SK_PER;
  const int col_len = idx < C ? max(0, min(R, i0 + SK_PER) - i0) : 0;          // rows [36 part, 36 part + 36) of a column
  for (int it = 0; it < iters; ++it) {
    {
      const float* row = S + idx * ld + part;
      float x[SK_PER];
      float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // four independent chains: two waves per SIMD do not
#pragma unroll                                                     // hide a 36-deep dependent max / add chain
      for (int t = 0; t < SK_PER; ++t) {
        const float sv = row[SK_PARTS * t] + v[part + SK_PARTS * t];  // unconditional: past the row's end the reads land in
        x[t] = t < row_len ? sv : -INFINITY;                          // the next row / the vectors behind S (or return 0)
        m4[t & 3] = fmaxf(m4[t & 3], x[t]);
      }
      const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) s4[t & 3] += t < row_len ? __expf(x[t] - mx) : 0.f;
      const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      const float lse = lse_merge4(mx, s);
      if (part == 0 && idx < R) u[idx] = log_mu[idx] - lse;
      for (int r = SK_MAIN + wv; r < R; r += SK_T / WAVE) {  // rows beyond the 128 that own four lanes: a wave each
        const float l2 = wave_lse(S + (size_t)r * ld, 1, v, C, lane);
        if (lane == 0) u[r] = log_mu[r] - l2;
      }
    }
    __syncthreads();
    {
      const float* col = S + (size_t)i0 * ld + idx;
      float x[SK_PER];
      float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) {
        const float sv = col[(size_t)t * ld] + u[i0 + t];
        x[t] = t < col_len ? sv : -INFINITY;
        m4[t & 3] = fmaxf(m4[t & 3], x[t]);
      }
      const float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) s4[t & 3] += t < col_len ? __expf(x[t] - mx) : 0.f;
      const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      const float lse = lse_merge4(mx, s);
      if (part == 0 && idx < C) v[idx] = log_nu[idx] - lse;
      for (int c = SK_MAIN + wv; c < C; c += SK_T / WAVE) {
        const float l2 = wave_lse(S + c, ld, u, R, lane);
        if (lane == 0) v[c] = log_nu[c] - l2;
      }
    }
    __syncthreads();
  }
  // scores + u + v - norm (:18, :65)
  for (int e = threadIdx.x; e < R * C; e += SK_T) {
    const int i = e / C, j = e % C;
    out[(int64_t)b * R * C + e] = ((S[i * ld + j] + u[i]) + v[j]) - norm;
  }
}

size_t sinkhorn_lds(int M, int N) {
  const int R = M + 1, C = N + 1, ld = sinkhorn_ld(C);
  return sizeof(float) * ((size_t)R * ld + 2 * R + 2 * C) + 64;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" int gr_sinkhorn(const float* scores, int64_t batch, int64_t m, int64_t n, const uint8_t* row_masks,
                           const uint8_t* col_masks, const float* alpha_dev, int num_iterations, float inf,
                           float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(batch >= 0 && m >= 1 && n >= 1 && num_iterations >= 0, "bad sizes");
  if (batch == 0) return GR_OK;
  GR_REQUIRE(scores && alpha_dev && out, "null argument");
  GR_REQUIRE(std::max(m, n) + 1 <= SK_MAXD, "sinkhorn: matrices larger than %d are not supported", SK_MAXD - 1);
  const size_t lds = sinkhorn_lds((int)m, (int)n);
  if (lds > 160 * 1024) {
    set_error("sinkhorn: a (%lld+1) x (%lld+1) matrix does not fit in LDS (%zu bytes)", (long long)m, (long long)n, lds);
    return GR_ERR_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sinkhorn_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  KernelTimer timer("sinkhorn", stream);
  hipLaunchKernelGGL(sinkhorn_kernel, dim3((unsigned)batch), dim3(SK_T), lds, stream, scores, (int)m, (int)n, row_masks,
                     col_masks, alpha_dev, num_iterations, inf, out);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
