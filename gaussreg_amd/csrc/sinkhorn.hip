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

constexpr int SS_MAX = 63;                   // sinkhorn_small_kernel: valid rows / columns (+ the dustbin = 64 lanes)
constexpr int SS_LD = 65;                    // row stride of its compacted matrix in LDS (conflict-free rows and columns)

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
__device__ __forceinline__ void sinkhorn_matrix(const int b, const float* __restrict__ scores, int M, int N,
                                                const uint8_t* __restrict__ row_masks,
                                                const uint8_t* __restrict__ col_masks,
                                                const float* __restrict__ alpha_p, int iters, float inf,
                                                float* __restrict__ out, int scaling_form, int drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = M + 1, C = N + 1;
  const int ld = sinkhorn_ld(C);
  float* S = reinterpret_cast<float*>(smem);
  float* u = S + (size_t)R * ld;
  float* v = u + R;
  float* log_mu = v + C;
  float* log_nu = log_mu + R;
  int* cnt = reinterpret_cast<int*>(log_nu + C);      // [0], [1]: valid rows / columns; [2]: the scaling form gave up
  // scaling form: row maxima, exp(v - c) grouped by part, exp(u + rmax - c'), u + rmax -- 16-byte aligned (read as float4)
  // (an offset from S, not a pointer rounded through an integer: that would turn every access into a flat load)
  float* rmax = S + (((size_t)R * ld + 2 * (size_t)R + 2 * (size_t)C + 4 + 3) & ~(size_t)3);
  float* Ep = rmax + SK_MAXD;
  float* Fv = Ep + SK_MAXD;
  const float alpha = alpha_p[0];
  const uint8_t* rm = row_masks ? row_masks + (int64_t)b * M : nullptr;
  const uint8_t* cm = col_masks ? col_masks + (int64_t)b * N : nullptr;
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  // valid row / column counts (learnable_sinkhorn.py:50-51)
  int nr = 0, nc = 0;
  for (int i = threadIdx.x; i < M; i += SK_T) nr += (!rm || rm[i]) ? 1 : 0;
  for (int j = threadIdx.x; j < N; j += SK_T) nc += (!cm || cm[j]) ? 1 : 0;
  if (nr) atomicAdd(&cnt[0], nr);
  if (nc) atomicAdd(&cnt[1], nc);
  __syncthreads();
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
  // ---- the same iteration in scaling form.  u = log_mu - logsumexp_j(S + v) is, for any constant c,
  //   u_i = log_mu_i - (rmax_i + c + log sum_j K_ij exp(v_j - c)),  K_ij = exp(S_ij - rmax_i)  (rmax_i = max_j S_ij),
  // and likewise v_j = log_nu_j - (c' + log sum_i K_ij exp(u_i + rmax_i - c')).  K does not change: every lane keeps its
  // 36 elements of the row slice and its 36 of the column slice in registers, and a half-iteration is 36 FMAs with the
  // 129 exponentials exp(v_j - c) / exp(u_i + rmax_i - c') (one per row / column, written by the lane that owns it) -- the
  // log-domain form evaluates 129 x 129 exponentials per half-iteration.  c, c' = the maxima of the PREVIOUS iteration's
  // vectors (any constant is exact; these keep the exponents near zero without a third barrier).  Masked rows / columns
  // (scores = -inf = -1e12) have K = 0 and keep u = v = 0: their outputs are the -1e12 stand-ins either way.  If a sum
  // leaves the normal range (score ranges beyond ~80) the matrix is redone in the log domain below.
  bool scaled = false;
  if (scaling_form && iters > 0) {
    float kr[SK_PER], kc[SK_PER];
    constexpr int NSIDE = (SK_MAXD + WAVE - 1) / WAVE;  // elements per lane of a row / column reduced by a whole wave
    float ks[NSIDE];                                    // wave 0: the side row's K, wave 1: the side column's
    bool bad = false;
    static_assert(SK_MAXD - SK_MAIN <= 2 * (SK_T / WAVE), "one side row per wave");
    const bool row_ok = idx < R && (idx >= M || !rm || rm[idx]);
    const bool col_ok = idx < C && (idx >= N || !cm || cm[idx]);
    // rows / columns beyond the 128 that own four lanes (the dustbins at the demo shape): side row r_side by wave
    // `wv` (even waves), side column c_side by the odd waves -- the two side reductions of an iteration sit on different waves
    const int r_side = SK_MAIN + wv / 2, c_side = SK_MAIN + wv / 2;
    const bool has_rside = (wv & 1) == 0 && r_side < R, has_cside = (wv & 1) == 1 && c_side < C;
    {
      const float* row = S + idx * ld + part;
      float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) {
        kr[t] = t < row_len ? row[SK_PARTS * t] : -INFINITY;
        m4[t & 3] = fmaxf(m4[t & 3], kr[t]);
      }
      float mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      mx = fmaxf(mx, quad_xor1(mx));
      mx = fmaxf(mx, quad_xor2(mx));
      // An entry of K that UNDERFLOWS drops a term of the column sums which its other factor (up to 1e30) could have made the
      // largest one: the range guard on the sums does not see that, so a matrix with such an entry on a live row and column
      // takes the log-domain iterations (rows spanning more than ~87: found with 30 % valid slots and scores ~ N(0, 40^2),
      // where the sums stayed in range and the result was off by 1.5e-3 of the scale).
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) {
        const int j = part + SK_PARTS * t;
        const bool live = t < row_len && row_ok;
        kr[t] = live ? __expf(kr[t] - mx) : 0.f;  // masked columns: exp(-1e12 - mx) = 0
        bad = bad || (live && kr[t] == 0.f && (j >= N || !cm || cm[j]));
      }
      if (part == 0 && idx < R) rmax[idx] = row_ok ? mx : 0.f;
      for (int r = SK_MAIN + wv; r < R; r += SK_T / WAVE) {  // (rows beyond M are never masked)
        float m = -INFINITY;
        for (int k = lane; k < C; k += WAVE) m = fmaxf(m, S[(size_t)r * ld + k]);
        m = wave_max_f32_dpp(m);
        if (lane == 0) rmax[r] = m;
      }
      // v = 0: exp(v) = 1 on the live columns
      for (int j = threadIdx.x; j < SK_MAXD; j += SK_T) {
        const bool live = j < C && (j >= N || !cm || cm[j]);
        Ep[(j & 3) * SK_PER + (j >> 2)] = live ? 1.f : 0.f;
        Fv[j] = 0.f;
      }
    }
    __syncthreads();
    {
      const float* col = S + (size_t)i0 * ld + idx;
#pragma unroll
      for (int t = 0; t < SK_PER; ++t) {
        const int i = i0 + t;
        const bool live = t < col_len && col_ok && (i >= M || !rm || rm[i]);
        kc[t] = live ? __expf(col[(size_t)t * ld] - rmax[min(i, R - 1)]) : 0.f;
      }
#pragma unroll
      for (int t = 0; t < NSIDE; ++t) {
        const int k = lane + t * WAVE;
        ks[t] = 0.f;
        if (has_rside && k < C) {
          ks[t] = __expf(S[(size_t)r_side * ld + k] - rmax[r_side]);  // masked columns give 0
          bad = bad || (ks[t] == 0.f && (k >= N || !cm || cm[k]));
        }
        if (has_cside && k < R && (k >= M || !rm || rm[k])) ks[t] = __expf(S[(size_t)k * ld + c_side] - rmax[k]);
      }
    }
    const float cw = norm;  // reference point of exp(u + rmax - cw): u + rmax = log_mu - log(row sum) stays near log_mu
    if (bad) cnt[3] = 1;
    __syncthreads();
    const int fast_iters = cnt[3] ? 0 : iters;  // (an underflowed entry: straight to the log domain)
    for (int it = 0; it < fast_iters; ++it) {
      {  // rows
        const float4* e4 = reinterpret_cast<const float4*>(Ep + part * SK_PER);
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t4 = 0; t4 < SK_PER / 4; ++t4) {
          const float4 e = e4[t4];
          s4[0] = fmaf(kr[4 * t4], e.x, s4[0]);
          s4[1] = fmaf(kr[4 * t4 + 1], e.y, s4[1]);
          s4[2] = fmaf(kr[4 * t4 + 2], e.z, s4[2]);
          s4[3] = fmaf(kr[4 * t4 + 3], e.w, s4[3]);
        }
        float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        sum += quad_xor1(sum);
        sum += quad_xor2(sum);
        if (part == 0 && row_ok) {
          bad = bad || !(sum > 1e-30f && sum < 1e30f);
          const float w = log_mu[idx] - __logf(sum);  // = u + rmax  (v's reference point is 0)
          u[idx] = w - rmax[idx];
          Fv[idx] = __expf(w - cw);
        }
        if (has_rside) {
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < NSIDE; ++t) {
            const int k = min(lane + t * WAVE, SK_MAXD - 1);
            acc = fmaf(ks[t], Ep[(k & 3) * SK_PER + (k >> 2)], acc);
          }
          acc = wave_sum_f32_dpp(acc);
          if (lane == 0) {
            bad = bad || !(acc > 1e-30f && acc < 1e30f);
            const float w = log_mu[r_side] - __logf(acc);
            u[r_side] = w - rmax[r_side];
            Fv[r_side] = __expf(w - cw);
          }
        }
      }
      __syncthreads();
      {  // columns
        const float4* f4 = reinterpret_cast<const float4*>(Fv + i0);
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t4 = 0; t4 < SK_PER / 4; ++t4) {
          const float4 f = f4[t4];
          s4[0] = fmaf(kc[4 * t4], f.x, s4[0]);
          s4[1] = fmaf(kc[4 * t4 + 1], f.y, s4[1]);
          s4[2] = fmaf(kc[4 * t4 + 2], f.z, s4[2]);
          s4[3] = fmaf(kc[4 * t4 + 3], f.w, s4[3]);
        }
        float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        sum += quad_xor1(sum);
        sum += quad_xor2(sum);
        // (Ep was last read by the row pass, before the barrier above; Fv is not written here: no hazard)
        if (part == 0 && col_ok) {
          bad = bad || !(sum > 1e-30f && sum < 1e30f);
          const float vn = log_nu[idx] - (cw + __logf(sum));
          v[idx] = vn;
          Ep[(idx & 3) * SK_PER + (idx >> 2)] = __expf(vn);
        }
        if (has_cside) {
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < NSIDE; ++t) acc = fmaf(ks[t], Fv[min(lane + t * WAVE, SK_MAXD - 1)], acc);
          acc = wave_sum_f32_dpp(acc);
          if (lane == 0) {
            bad = bad || !(acc > 1e-30f && acc < 1e30f);
            const float vn = log_nu[c_side] - (cw + __logf(acc));
            v[c_side] = vn;
            Ep[(c_side & 3) * SK_PER + (c_side >> 2)] = __expf(vn);
          }
        }
      }
      __syncthreads();
    }
    if (bad) cnt[2] = 1;
    __syncthreads();
    scaled = cnt[2] == 0 && cnt[3] == 0;
    if (!scaled) {  // start over in the log domain
      for (int i = threadIdx.x; i < R; i += SK_T) u[i] = 0.f;
      for (int j = threadIdx.x; j < C; j += SK_T) v[j] = 0.f;
      __syncthreads();
    }
  }
  for (int it = 0; it < (scaled ? 0 : iters); ++it) {
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
  // scores + u + v - norm (:18, :65); drop: without the dustbin row and column (what model.py:197-198 slices off)
  const int OR = drop ? M : R, OC = drop ? N : C;
  for (int e = threadIdx.x; e < OR * OC; e += SK_T) {
    const int i = e / OC, j = e % OC;
    out[(int64_t)b * OR * OC + e] = ((S[i * ld + j] + u[i]) + v[j]) - norm;
  }
}

// One workgroup per matrix -- or, behind sinkhorn_small_kernel, a few hundred workgroups that walk the list of the matrices
// that kernel left (more than 63 valid rows or columns, or rejected by its range guard): worklist[0] = their number,
// worklist[1..] = their indices.  (A grid of one 70 KB workgroup per matrix that exits at once where nothing is to do still
// took 0.2 ms per 4 096 matrices to dispatch.)
__global__ __launch_bounds__(SK_T) void sinkhorn_kernel(const float* __restrict__ scores, int M, int N,
                                                        const uint8_t* __restrict__ row_masks,
                                                        const uint8_t* __restrict__ col_masks,
                                                        const float* __restrict__ alpha_p, int iters, float inf,
                                                        float* __restrict__ out, int scaling_form,
                                                        const int32_t* __restrict__ worklist, int batch, int drop) {
  const int n_items = worklist ? worklist[0] : batch;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    sinkhorn_matrix(worklist ? worklist[1 + item] : item, scores, M, N, row_masks, col_masks, alpha_p, iters, inf, out,
                    scaling_form, drop);
    __syncthreads();  // the next matrix reuses the LDS image
  }
}

// ---------------------------------------------------------------- the same transport for SMALL problems: one wave per matrix
// A patch of GaussReg's fine matching has 128 slots per side, but a superpoint owns ~30 points: three quarters of the rows and
// columns are masked, and the masked ones take no part in the iteration (K = 0, u = v = 0).  When at most 63 rows and 63
// columns are valid, the valid (rows + dustbin) x (columns + dustbin) problem fits ONE wave: lane l owns compacted row l AND
// compacted column l, keeps both (64 + 64 values of K = exp(S - rowmax)) in registers, and a half-iteration is 64 FMAs against
// a vector read from LDS as broadcast float4 -- no workgroup barrier, no work on masked entries (the 512-thread kernel above
// spends 129 x 129 FMAs per half-iteration whatever the masks say: 8.3 ms per 16 384 patches of the pair path).  Same
// scaling-form arithmetic and the same range guard; a matrix the guard rejects, like one that is too large, is appended to a work
// list for the kernel above, which is launched behind this one with a few hundred workgroups that walk that list.  Masked entries of the output are
// ((-inf + u_i) + v_j) - norm with u = v = 0 on masked rows / columns, as above.

__global__ __launch_bounds__(WAVE) void sinkhorn_small_kernel(const float* __restrict__ scores, int M, int N,
                                                              const uint8_t* __restrict__ row_masks,
                                                              const uint8_t* __restrict__ col_masks,
                                                              const float* __restrict__ alpha_p, int iters, float inf,
                                                              float* __restrict__ out, int32_t* __restrict__ worklist, int drop) {
  __shared__ float S[WAVE * SS_LD];
  __shared__ __attribute__((aligned(16))) float Ev[WAVE];
  __shared__ __attribute__((aligned(16))) float Fu[WAVE];
  __shared__ float uu[WAVE], vv[WAVE], rmaxv[WAVE];
  __shared__ short rlist[WAVE], clist[WAVE], rinv[SK_MAXD], cinv[SK_MAXD];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int R = M + 1, C = N + 1;
  const uint8_t* rm = row_masks ? row_masks + (int64_t)b * M : nullptr;
  const uint8_t* cm = col_masks ? col_masks + (int64_t)b * N : nullptr;
  const float alpha = alpha_p[0];
  // compacted lists of the valid rows / columns (in index order), and their inverses (-1: masked)
  int nr = 0, nc = 0;
  for (int base = 0; base < M; base += WAVE) {
    const int i = base + lane;
    const bool ok = i < M && (!rm || rm[i]);
    const unsigned long long m = __ballot(ok);
    const int pos = nr + (int)__popcll(m & ((1ull << lane) - 1ull));
    if (ok && pos < WAVE) rlist[pos] = (short)i;
    if (i < M) rinv[i] = ok && pos < WAVE ? (short)pos : (short)-1;
    nr += (int)__popcll(m);
  }
  for (int base = 0; base < N; base += WAVE) {
    const int j = base + lane;
    const bool ok = j < N && (!cm || cm[j]);
    const unsigned long long m = __ballot(ok);
    const int pos = nc + (int)__popcll(m & ((1ull << lane) - 1ull));
    if (ok && pos < WAVE) clist[pos] = (short)j;
    if (j < N) cinv[j] = ok && pos < WAVE ? (short)pos : (short)-1;
    nc += (int)__popcll(m);
  }
  if (nr > SS_MAX || nc > SS_MAX) {  // the 512-thread kernel takes this matrix
    if (lane == 0) worklist[1 + atomicAdd(&worklist[0], 1)] = b;
    return;
  }
  if (lane == 0) rinv[M] = (short)nr, cinv[N] = (short)nc;  // the dustbins close the compacted lists
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // compacted padded scores: lanes = columns, one row per step (coalesced when the valid columns are a prefix, as the
  // neighbour lists of point_to_node_partition make them)
  {
    const int gj = lane < nc ? clist[lane] : N;
#pragma unroll 4
    for (int r = 0; r <= nr; ++r) {
      const int gi = r < nr ? rlist[r] : M;
      if (lane <= nc) S[r * SS_LD + lane] = (gi < M && gj < N) ? scores[((int64_t)b * M + gi) * N + gj] : alpha;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float nvr = (float)nr, nvc = (float)nc;
  const float norm = -logf(nvr + nvc);                                                    // learnable_sinkhorn.py:52
  const bool rowl = lane <= nr, coll = lane <= nc;
  const float log_mu = lane < nr ? norm : logf(nvc) + norm;                               // :54-56 (lane nr: the dustbin row)
  const float log_nu = lane < nc ? norm : logf(nvr) + norm;                               // :59-61
  float kr[WAVE], kc[WAVE];
  float rmax = -INFINITY;
#pragma unroll
  for (int t = 0; t < WAVE; ++t) {
    const float x = S[lane * SS_LD + t];  // (unconditional: past the row's end the value is dropped)
    kr[t] = (rowl && t <= nc) ? x : -INFINITY;
    rmax = fmaxf(rmax, kr[t]);
  }
  if (!rowl) rmax = 0.f;
  // An entry of K that underflows drops a term of the sums that the other factor (up to 1e30) could have made the largest:
  // the range guard on the sums does not see that, so a matrix whose rows span more than ~87 takes the log-domain kernel.
  bool bad = false;
#pragma unroll
  for (int t = 0; t < WAVE; ++t) {
    const bool live = rowl && t <= nc;
    kr[t] = live ? __expf(kr[t] - rmax) : 0.f;
    bad = bad || (live && kr[t] == 0.f);
  }
  rmaxv[lane] = rmax;
  Ev[lane] = coll ? 1.f : 0.f;  // v = 0
  Fu[lane] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < WAVE; ++t) {
    const float x = S[t * SS_LD + lane];
    kc[t] = (coll && t <= nr) ? __expf(x - rmaxv[t]) : 0.f;
  }
  const float cw = norm;
  float u = 0.f, v = 0.f;
  for (int it = 0; it < (__any(bad) ? 0 : iters); ++it) {
    {  // rows: lane l = compacted row l
      const float4* e4 = reinterpret_cast<const float4*>(Ev);
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t4 = 0; t4 < WAVE / 4; ++t4) {
        const float4 e = e4[t4];
        s4[0] = fmaf(kr[4 * t4], e.x, s4[0]);
        s4[1] = fmaf(kr[4 * t4 + 1], e.y, s4[1]);
        s4[2] = fmaf(kr[4 * t4 + 2], e.z, s4[2]);
        s4[3] = fmaf(kr[4 * t4 + 3], e.w, s4[3]);
      }
      const float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      if (rowl) {
        bad = bad || !(sum > 1e-30f && sum < 1e30f);
        const float w = log_mu - __logf(sum);  // = u + rmax
        u = w - rmax;
        Fu[lane] = __expf(w - cw);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {  // columns: lane l = compacted column l
      const float4* f4 = reinterpret_cast<const float4*>(Fu);
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t4 = 0; t4 < WAVE / 4; ++t4) {
        const float4 f = f4[t4];
        s4[0] = fmaf(kc[4 * t4], f.x, s4[0]);
        s4[1] = fmaf(kc[4 * t4 + 1], f.y, s4[1]);
        s4[2] = fmaf(kc[4 * t4 + 2], f.z, s4[2]);
        s4[3] = fmaf(kc[4 * t4 + 3], f.w, s4[3]);
      }
      const float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      if (coll) {
        bad = bad || !(sum > 1e-30f && sum < 1e30f);
        v = log_nu - (cw + __logf(sum));
        Ev[lane] = __expf(v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (__any(bad)) {  // out of the normal range: the log-domain iterations of the 512-thread kernel redo this matrix
    if (lane == 0) worklist[1 + atomicAdd(&worklist[0], 1)] = b;
    return;
  }
  uu[lane] = rowl ? u : 0.f;
  vv[lane] = coll ? v : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // scores + u + v - norm (:18, :65) for the whole padded matrix
  const int OR = drop ? M : R, OC = drop ? N : C;  // drop: without the dustbin row and column (model.py:197-198)
  float* o = out + (int64_t)b * OR * OC;
  for (int i = 0; i < OR; ++i) {
    const int ri = rinv[i];  // uniform
    const float ui = ri >= 0 ? uu[ri] : 0.f;
    for (int j = lane; j < OC; j += WAVE) {
      const int cj = cinv[j];
      const float sv = (ri >= 0 && cj >= 0) ? S[ri * SS_LD + cj] : -inf;
      o[(int64_t)i * OC + j] = ((sv + ui) + (cj >= 0 ? vv[cj] : 0.f)) - norm;
    }
  }
}

size_t sinkhorn_lds(int M, int N) {
  const int R = M + 1, C = N + 1, ld = sinkhorn_ld(C);
  return sizeof(float) * ((size_t)R * ld + 2 * R + 2 * C + 4 * SK_MAXD) + 96;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_sinkhorn_workspace_bytes(int64_t batch) {
  return sizeof(int32_t) * (size_t)(std::max<int64_t>(batch, 0) + 1) + 256;
}

extern "C" int gr_sinkhorn(const float* scores, int64_t batch, int64_t m, int64_t n, const uint8_t* row_masks,
                           const uint8_t* col_masks, const float* alpha_dev, int num_iterations, float inf,
                           int drop_dustbin, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
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
  const int scaling_form = 1;  // (0: every iteration as logsumexp -- what the kernel itself falls back to, matrix by matrix)
  // matrices with at most 63 valid rows and columns: one wave each
  const int small = num_iterations > 0 && (row_masks || col_masks || std::max(m, n) <= SS_MAX);
  KernelTimer timer("sinkhorn", stream);
  if (small) {
    // [0] = number of matrices left for the 512-thread kernel, then their indices (caller's workspace: no allocation
    // inside the call, so it can be captured into a graph and nothing can leak on an error path)
    if (!workspace || workspace_bytes < gr_sinkhorn_workspace_bytes(batch)) {
      set_error("sinkhorn: workspace too small (%zu < %zu bytes)", workspace_bytes, gr_sinkhorn_workspace_bytes(batch));
      return GR_ERR_WORKSPACE;
    }
    int32_t* worklist = reinterpret_cast<int32_t*>(workspace);
    GR_HIP(hipMemsetAsync(worklist, 0, sizeof(int32_t), stream));
    hipLaunchKernelGGL(sinkhorn_small_kernel, dim3((unsigned)batch), dim3(WAVE), 0, stream, scores, (int)m, (int)n, row_masks,
                       col_masks, alpha_dev, num_iterations, inf, out, worklist, drop_dustbin ? 1 : 0);
    hipLaunchKernelGGL(sinkhorn_kernel, dim3((unsigned)std::min<int64_t>(batch, 512)), dim3(SK_T), lds, stream, scores, (int)m,
                       (int)n, row_masks, col_masks, alpha_dev, num_iterations, inf, out, scaling_form, worklist, (int)batch, drop_dustbin ? 1 : 0);
    GR_LAUNCH_CHECK();
    return GR_OK;
  }
  hipLaunchKernelGGL(sinkhorn_kernel, dim3((unsigned)batch), dim3(SK_T), lds, stream, scores, (int)m, (int)n, row_masks,
                     col_masks, alpha_dev, num_iterations, inf, out, scaling_form, (const int32_t*)nullptr, (int)batch, drop_dustbin ? 1 : 0);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
