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

constexpr int SK_PARTS = 4;   // threads per row / column
constexpr int SK_T = 576;     // >= SK_PARTS * (max(M,N)+1) for the demo shape (4 * 129 = 516), 9 waves
constexpr int SK_PER = (SK_T / SK_PARTS + SK_PARTS - 1) / SK_PARTS;  // elements of a row / column one thread reduces (<= 36)

__device__ __forceinline__ float lse_finish(float mx, float s) { return logf(s) + mx; }

// Thread (part p, index i) reduces a contiguous quarter of row / column i to a (max, sum-exp) partial;
// part 0 merges the four partials in fixed order.  Row stride is odd, so lanes that differ in i hit
// different banks in both passes.
__global__ __launch_bounds__(SK_T) void sinkhorn_kernel(const float* __restrict__ scores, int M, int N,
                                                        const uint8_t* __restrict__ row_masks,
                                                        const uint8_t* __restrict__ col_masks,
                                                        const float* __restrict__ alpha_p, int iters, float inf,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = M + 1, C = N + 1;
  const int ld = C | 1;
  const int mxd = max(R, C);
  float* S = reinterpret_cast<float*>(smem);
  float* u = S + (size_t)R * ld;
  float* v = u + R;
  float* log_mu = v + C;
  float* log_nu = log_mu + R;
  float* pm = log_nu + C;                 // [SK_PARTS][mxd] partial max
  float* ps = pm + SK_PARTS * mxd;        // [SK_PARTS][mxd] partial sum
  int* cnt = reinterpret_cast<int*>(ps + SK_PARTS * mxd);
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
  const int part = threadIdx.x / mxd, idx = threadIdx.x % mxd;  // part >= SK_PARTS: idle thread
  const bool active = part < SK_PARTS;
  const int cper = (C + SK_PARTS - 1) / SK_PARTS, rper = (R + SK_PARTS - 1) / SK_PARTS;
  for (int it = 0; it < iters; ++it) {
    if (active && idx < R) {
      const float* row = S + idx * ld;
      // the thread's slice stays in registers between the max pass and the sum pass (one trip through LDS per element)
      const int j0 = part * cper, len = min(C, j0 + cper) - j0;
      float x[SK_PER];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) {
        x[t] = t < len ? row[j0 + t] + v[j0 + t] : -INFINITY;
        mx = fmaxf(mx, x[t]);
      }
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) s += t < len ? __expf(x[t] - mx) : 0.f;
      pm[part * mxd + idx] = mx;
      ps[part * mxd + idx] = s;
    }
    __syncthreads();
    if (part == 0 && idx < R) {
      float mx = pm[idx];
#pragma unroll
      for (int p = 1; p < SK_PARTS; ++p) mx = fmaxf(mx, pm[p * mxd + idx]);
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < SK_PARTS; ++p) {
        const float m_p = pm[p * mxd + idx];
        if (m_p > -INFINITY) s += ps[p * mxd + idx] * expf(m_p - mx);  // empty part: (-inf, 0)
      }
      u[idx] = log_mu[idx] - lse_finish(mx, s);
    }
    __syncthreads();
    if (active && idx < C) {
      const int i0 = part * rper, len = min(R, i0 + rper) - i0;
      float x[SK_PER];
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) {
        x[t] = t < len ? S[(i0 + t) * ld + idx] + u[i0 + t] : -INFINITY;
        mx = fmaxf(mx, x[t]);
      }
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) s += t < len ? __expf(x[t] - mx) : 0.f;
      pm[part * mxd + idx] = mx;
      ps[part * mxd + idx] = s;
    }
    __syncthreads();
    if (part == 0 && idx < C) {
      float mx = pm[idx];
#pragma unroll
      for (int p = 1; p < SK_PARTS; ++p) mx = fmaxf(mx, pm[p * mxd + idx]);
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < SK_PARTS; ++p) {
        const float m_p = pm[p * mxd + idx];
        if (m_p > -INFINITY) s += ps[p * mxd + idx] * expf(m_p - mx);
      }
      v[idx] = log_nu[idx] - lse_finish(mx, s);
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
  const int R = M + 1, C = N + 1, ld = C | 1, mxd = R > C ? R : C;
  return sizeof(float) * ((size_t)R * ld + 2 * R + 2 * C + 2 * SK_PARTS * mxd) + 64;
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
  GR_REQUIRE(SK_PARTS * (std::max(m, n) + 1) <= SK_T, "sinkhorn: matrices larger than %d are not supported",
             SK_T / SK_PARTS - 1);
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
