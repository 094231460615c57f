// KPConv forward ("next" row) without the (M,H,C) / (M,H,K) / (K,M,C) temporaries.
//
// Replaces  geotransformer/modules/kpconv/kpconv.py:90-120  (and functional.py:6-22, 54-67 for the two
// pooling helpers).  The reference gathers neighbor_feats (M,H,C) -- 1.2 GB at stage 2 of the demo
// pyramid -- and runs two batched matmuls over it.  Here:
//   rowflag   flag[n] = (sum_c feats[n,c] > 0)                       (kpconv.py:112-113, once per call)
//   gather    one workgroup per query: kernel-point influences w[h,k] = max(1 - |y_h - kp_k| / sigma, 0)
//             computed once into LDS, then every thread owns a channel and accumulates the K weighted sums
//             over the neighbours with coalesced feature-row reads            -> WF (M, K*Cin)
//   gemm      out = WF (M x K*Cin) . W (K*Cin x Cout) on fp32 MFMA 32x32x2, epilogue / neighbor_num + bias
#include <algorithm>

#include "common.hpp"

namespace gr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KP_MAX = 16;   // kernel points (config.py:84 kernel_size = 15)
constexpr int KP_HMAX = 256; // neighbours per query staged at once

__global__ __launch_bounds__(256) void rowflag_kernel(const float* __restrict__ f, int n, int C,
                                                      uint8_t* __restrict__ flag) {
  const int row = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  if (row >= n) return;
  float s = 0.f;
  for (int c = lane; c < C; c += WAVE) s += f[(int64_t)row * C + c];
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) s += __shfl_xor(s, d, WAVE);
  if (lane == 0) flag[row] = s > 0.0f ? 1 : 0;
}

template <int T>
__global__ __launch_bounds__(T) void kp_gather_kernel(
    const float* __restrict__ s_feats, const float* __restrict__ q_points, const float* __restrict__ s_points,
    const int64_t* __restrict__ nbr, int N, int H, int Cin, int K, const float* __restrict__ kpts, float sigma,
    float inf, const uint8_t* __restrict__ flag, float* __restrict__ WF, float* __restrict__ inv_num) {
  // sized per launch for min(H, KP_HMAX) neighbours: a fixed 256-row table (17 KB) capped the kernel at 9 one-wave
  // workgroups per CU, and every workgroup is a chain of three dependent gathers
  extern __shared__ __attribute__((aligned(16))) float kp_smem[];
  const int hcap = min(H, KP_HMAX);
  float (*s_w)[KP_MAX + 1] = reinterpret_cast<float (*)[KP_MAX + 1]>(kp_smem);
  int* s_idx = reinterpret_cast<int*>(kp_smem + (size_t)hcap * (KP_MAX + 1));
  __shared__ int s_cnt;
  const int m = blockIdx.x;
  const float qx = q_points[3 * (int64_t)m], qy = q_points[3 * (int64_t)m + 1], qz = q_points[3 * (int64_t)m + 2];
  float acc[KP_MAX];
#pragma unroll
  for (int k = 0; k < KP_MAX; ++k) acc[k] = 0.f;
  if (threadIdx.x == 0) s_cnt = 0;
  for (int h0 = 0; h0 < H; h0 += KP_HMAX) {
    const int hn = min(KP_HMAX, H - h0);
    __syncthreads();
    int local = 0;
    for (int h = threadIdx.x; h < hn; h += T) {
      const int64_t idx = nbr[(int64_t)m * H + h0 + h];
      const bool pad = idx >= N || idx < 0;
      s_idx[h] = pad ? -1 : (int)idx;
      // kpconv.py:90-92: shadow support at +inf, neighbours centred on the query
      const float nx = (pad ? inf : s_points[3 * idx]) - qx;
      const float ny = (pad ? inf : s_points[3 * idx + 1]) - qy;
      const float nz = (pad ? inf : s_points[3 * idx + 2]) - qz;
      for (int k = 0; k < K; ++k) {
        const float dx = nx - kpts[3 * k], dy = ny - kpts[3 * k + 1], dz = nz - kpts[3 * k + 2];
        const float sq = (dx * dx + dy * dy) + dz * dz;                   // :97
        s_w[h][k] = fmaxf(1.0f - sqrtf(sq) / sigma, 0.0f);                // :98
      }
      local += (!pad && flag[idx]) ? 1 : 0;                               // :112-113
    }
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    for (int c = threadIdx.x; c < Cin; c += T) {
      // a thread owns channel c only when Cin <= T; otherwise it flushes per c below
      float a[KP_MAX];
#pragma unroll
      for (int k = 0; k < KP_MAX; ++k) a[k] = 0.f;
      // four neighbour rows requested before the first one is consumed (the loop is a chain of gathers otherwise);
      // a shadow neighbour contributes an exact zero either way (:103), so it is loaded as 0 instead of skipped
      int h = 0;
      for (; h + 4 <= hn; h += 4) {
        float fv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = s_idx[h + u];
          fv[u] = idx < 0 ? 0.f : s_feats[(int64_t)idx * Cin + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s_idx[h + u] >= 0) {
#pragma unroll
            for (int k = 0; k < KP_MAX; ++k) a[k] = fmaf(s_w[h + u][k], fv[u], a[k]);  // :105 (M,K,H) x (M,H,C)
          }
      }
      for (; h < hn; ++h) {
        const int idx = s_idx[h];
        if (idx < 0) continue;                                            // zero shadow feature (:103)
        const float fv = s_feats[(int64_t)idx * Cin + c];
#pragma unroll
        for (int k = 0; k < KP_MAX; ++k) a[k] = fmaf(s_w[h][k], fv, a[k]);
      }
      if (Cin <= T) {
#pragma unroll
        for (int k = 0; k < KP_MAX; ++k) acc[k] += a[k];
      } else {
        float* dst = WF + (int64_t)m * K * Cin + c;
        for (int k = 0; k < K; ++k) dst[(int64_t)k * Cin] = (h0 == 0 ? 0.f : dst[(int64_t)k * Cin]) + a[k];
      }
    }
  }
  if (Cin <= T && threadIdx.x < Cin) {
    float* dst = WF + (int64_t)m * K * Cin + threadIdx.x;
    for (int k = 0; k < K; ++k) dst[(int64_t)k * Cin] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) inv_num[m] = (float)max(s_cnt, 1);                // :114 max(neighbor_num, 1)
}

// The same on the matrix cores, for Cin a multiple of 16: WF[m] (K x Cin) = A^T (K x H) . F (H x Cin) with A[h][k] the kernel-point
// influence of neighbour h and F its feature row -- a tiny GEMM per query, one WAVE per query.  v_mfma_f32_16x16x4f32 takes
// A as (16 kernel points x 4 neighbours) with lane l supplying A[l % 16][l / 16] and B as (4 neighbours x 16 channels) with
// lane l supplying B[l / 16][l % 16]: every lane computes ONE influence per step (its neighbour, its kernel point; nothing
// staged in LDS) and loads one feature word per 16-channel tile; shadow neighbours contribute an influence of exactly 0 and
// a feature of 0.  Cin / 16 MFMAs per four neighbours replace 15 x 4 FMAs per lane: the VALU form above is bound by those
// FMAs (0.88 ms per layer at 960 k queries x 64 channels), not by the gathers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KP_STEPS = 16;  // steps of four neighbours the MFMA gather holds indices for: H <= 64

// Channel <-> (tile, lane) mapping: a lane loads VEC = min(4, NT) consecutive channels of its neighbour's row in one request
// (16 lanes x 16 bytes = a whole 256-byte row at Cin = 64) and component c of that vector is its B value for tile c of the
// group -- a tile is the strided channel set {VEC j + c}, which is as good a 16-channel tile as a contiguous one; the D
// registers then leave as the same vectors.
template <int NT>  // 16-channel tiles: Cin = 16 NT
__global__ __launch_bounds__(256) void kp_gather_mfma_kernel(
    const float* __restrict__ s_feats, const float* __restrict__ q_points, const float* __restrict__ s_points,
    const int64_t* __restrict__ nbr, int N, int M, int H, int K, const float* __restrict__ kpts, float sigma, float inf,
    const uint8_t* __restrict__ flag, float* __restrict__ WF, float* __restrict__ inv_num) {
  constexpr int Cin = 16 * NT;
  constexpr int VEC = NT < 4 ? NT : 4;  // channels per load (1, 2 or 4)
  constexpr int NG = NT / VEC;          // loads per neighbour: groups of 16 VEC channels
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & (WAVE - 1);
  const int m = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (m >= M) return;  // wave-uniform; the kernel has no barrier
  const int kk = lane & 15, hq = lane >> 4;  // this lane's kernel point / neighbour of the step (A), channel column (B)
  const float qx = q_points[3 * (int64_t)m], qy = q_points[3 * (int64_t)m + 1], qz = q_points[3 * (int64_t)m + 2];
  const bool kreal = kk < K;
  const float kx = kreal ? kpts[3 * kk] : 0.f, ky = kreal ? kpts[3 * kk + 1] : 0.f, kz = kreal ? kpts[3 * kk + 2] : 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  const int64_t* row = nbr + (int64_t)m * H;
  // A step is a chain index -> (support point, feature row) -> MFMAs.  All indices of the row are requested first, and the
  // loads of step s + 1 are issued before the MFMAs of step s, so the wave waits for one memory round trip per row plus
  // one per step that the arithmetic of the step before does not cover (H <= 4 KP_STEPS; the host checks).
  int idxs[KP_STEPS];
#pragma unroll
  for (int sidx = 0; sidx < KP_STEPS; ++sidx) {
    const int h = 4 * sidx + hq;
    const int64_t v = h < H ? row[h] : -1;
    idxs[sidx] = (v >= N || v < 0) ? -1 : (int)v;
  }
  const int nsteps = (H + 3) / 4;
  float nxn, nyn, nzn;
  vec_t bn[NG];
  int fln;
  auto fetch = [&](int sidx) {
    const int id = idxs[sidx];
    const int64_t si = id < 0 ? 0 : id;
    nxn = s_points[3 * si], nyn = s_points[3 * si + 1], nzn = s_points[3 * si + 2];
    fln = flag[si];
    const vec_t* frow = reinterpret_cast<const vec_t*>(s_feats + si * Cin) + kk;
#pragma unroll
    for (int g = 0; g < NG; ++g) bn[g] = frow[16 * g];                    // unconditional loads, zeros selected below (:103)
  };
  fetch(0);
#pragma unroll
  for (int sidx = 0; sidx < KP_STEPS; ++sidx) {
    if (sidx < nsteps) {  // wave-uniform
      const bool pad = idxs[sidx] < 0;
      const float px_ = nxn, py_ = nyn, pz_ = nzn;
      const int fl = fln;
      vec_t b[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g] = bn[g];
      if (sidx + 1 < KP_STEPS && sidx + 1 < nsteps) fetch(sidx + 1);
      // kpconv.py:90-98: shadow support at +inf, neighbours centred on the query, influence max(1 - |.|/sigma, 0)
      const float nx = (pad ? inf : px_) - qx, ny = (pad ? inf : py_) - qy, nz = (pad ? inf : pz_) - qz;
      const float dx = nx - kx, dy = ny - ky, dz = nz - kz;
      const float sq = (dx * dx + dy * dy) + dz * dz;
      const float a = (kreal && !pad) ? fmaxf(1.0f - sqrtf(sq) / sigma, 0.0f) : 0.0f;
      cnt += (kk == 0 && !pad && fl) ? 1 : 0;                             // :112-113, one lane per neighbour
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          const float bv = b[g][c];
          acc[g * VEC + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pad ? 0.0f : bv, acc[g * VEC + c], 0, 0, 0);
        }
    }
  }
  // D[i][j]: lane l holds rows i = 4 (l / 16) + r, column j = l % 16  ->  WF[m][kernel point i][channels 16 VEC g + VEC j + c]
  vec_t* dst = reinterpret_cast<vec_t*>(WF + (int64_t)m * K * Cin) + kk;
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * hq + r;
      vec_t o;
#pragma unroll
      for (int c = 0; c < VEC; ++c) o[c] = acc[g * VEC + c][r];
      // streaming store: WF is read back once, by the product launch, long after it has left the caches; kept out of them
      // the neighbours' feature rows (re-gathered ~H times) stay (14 KPConv calls of a 32-pair batch 81.2 -> 79.4 ms;
      // streaming LOADS of WF in the product measured slower: 84 ms)
      if (i < K) __builtin_nontemporal_store(o, dst + (int64_t)i * (Cin / VEC) + 16 * g);
    }
  cnt = wave_sum_i32_dpp(cnt);
  if (lane == 0) inv_num[m] = (float)max(cnt, 1);                         // :114 max(neighbor_num, 1)
}

constexpr int GT = 64, GK = 32, GLD = GK + 1;

// C (M x N) = A (M x Kd, row-major) . B (Kd x N, row-major);  out = C / den[m] + bias[n]
__global__ __launch_bounds__(256) void gemm_nn_kernel(const float* __restrict__ A, const float* __restrict__ B, int M,
                                                      int N, int Kd, const float* __restrict__ den,
                                                      const float* __restrict__ bias, float* __restrict__ out) {
  __shared__ float sa[GT][GLD];
  __shared__ float sb[GT][GLD];
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < Kd; k0 += GK) {
    for (int e = tid; e < GT * GK; e += 256) {
      const int r = e / GK, k = e % GK;   // A tile: coalesced along k
      const int gi = i0 + r, gk = k0 + k;
      sa[r][k] = (gi < M && gk < Kd) ? A[(int64_t)gi * Kd + gk] : 0.f;
      const int kk = e / GT, j = e % GT;  // B tile: coalesced along j, stored transposed
      const int gj = j0 + j, gkb = k0 + kk;
      sb[j][kk] = (gj < N && gkb < Kd) ? B[(int64_t)gkb * N + gj] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; k += 2) {
      const float a = sa[wi + (lane & 31)][k + (lane >> 5)];
      const float b = sb[wj + (lane & 31)][k + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gi = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gj = j0 + wj + (lane & 31);
    if (gi < M && gj < N) {
      float v = acc[r];
      if (den) v = v / den[gi];          // kpconv.py:115
      if (bias) v = v + bias[gj];        // :118-119
      out[(int64_t)gi * N + gj] = v;
    }
  }
}

// Larger tile for the layers that dominate the backbone (M >= 128, N >= 128): 128 x 128 x 16 block tile,
// 64 x 64 per wave = 2 x 2 MFMA 32x32x2 accumulators (A and B fragments are each reused twice), LDS
// double-buffered so the global loads of slab k+1 overlap the 32 MFMAs of slab k.
constexpr int BT = 128, BK = 16, BLD = BK + 1;

// BM = rows per workgroup: 128 (2 x 2 waves of 64 x 64) or 64 (2 x 2 waves of 32 x 64) -- the smaller tile doubles the
// number of workgroups when M is only ~10^4 rows and the K loop (15 * Cin long, serial inside a workgroup) would
// otherwise leave the matrix pipes of half the chip waiting on one wave per SIMD.
// ALIGNED (Kd a multiple of the 16-wide slab, N a multiple of 4, both matrices 16-byte aligned): every fetch is one
// unconditional float4 load on a clamped row / column -- rows and columns past the edge compute values nobody stores.  Behind
// the bounds checks of the general path the compiler keeps each 4-byte load behind the previous one's use: sixteen memory
// round trips per slab instead of one (the same step took the distance kernel from 56 to 92 TFLOP/s).
template <int BM, int BN, bool ALIGNED = false>
__global__ __launch_bounds__(256) void gemm_nn_big_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          int M, int N, int Kd, const float* __restrict__ den,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
  constexpr int WM = BM / 2, WN = BN / 2;    // rows / columns per wave (2 x 2 waves)
  constexpr int TA = WM / 32, TB = WN / 32;  // 32 x 32 MFMA tiles per wave
  __shared__ float sa[2][BM][BLD];
  __shared__ float sb[2][BN][BLD];
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 1) * WM, wj = (w & 1) * WN;
  f32x16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // staging: A tile BM rows x 16 k (coalesced along k: 16 threads per row), B tile 16 k x BN cols
  constexpr int NA = BM * BK / 256, NBv = BN * BK / 256;
  float ra[NA], rb[NBv];
  // ALIGNED staging: A -- thread (row tid / 4 + 64 u, four consecutive k); B -- thread (k, four consecutive columns)
  constexpr int VA = BM / 64, VB = BN / 64, BQ = BN / 4;  // float4 per thread and slab; float4 per B row of the tile
  const float* arow[VA];
  const float* bcol[VB];
  if (ALIGNED) {
#pragma unroll
    for (int u = 0; u < VA; ++u) arow[u] = A + (int64_t)min(i0 + (tid >> 2) + 64 * u, M - 1) * Kd + (tid & 3) * 4;
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int e = tid + u * 256;
      bcol[u] = B + (int64_t)(e / BQ) * N + min(j0 + (e % BQ) * 4, N - 4);
    }
  }
  auto load = [&](int k0) {
    if (ALIGNED) {
#pragma unroll
      for (int u = 0; u < VA; ++u) {
        const float4 t = *reinterpret_cast<const float4*>(arow[u] + k0);
        ra[4 * u] = t.x, ra[4 * u + 1] = t.y, ra[4 * u + 2] = t.z, ra[4 * u + 3] = t.w;
      }
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        const float4 t = *reinterpret_cast<const float4*>(bcol[u] + (int64_t)k0 * N);
        rb[4 * u] = t.x, rb[4 * u + 1] = t.y, rb[4 * u + 2] = t.z, rb[4 * u + 3] = t.w;
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + u * 256;
      const int r = e / BK, k = e % BK;
      const int gi = i0 + r, gk = k0 + k;
      ra[u] = (gi < M && gk < Kd) ? A[(int64_t)gi * Kd + gk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NBv; ++u) {
      const int e = tid + u * 256;
      const int kk = e / BN, j = e % BN;
      const int gj = j0 + j, gkb = k0 + kk;
      rb[u] = (gj < N && gkb < Kd) ? B[(int64_t)gkb * N + gj] : 0.f;
    }
  };
  auto store = [&](int buf) {
    if (ALIGNED) {
#pragma unroll
      for (int u = 0; u < VA; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) sa[buf][(tid >> 2) + 64 * u][(tid & 3) * 4 + q] = ra[4 * u + q];
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        const int e = tid + u * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) sb[buf][(e % BQ) * 4 + q][e / BQ] = rb[4 * u + q];
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + u * 256;
      sa[buf][e / BK][e % BK] = ra[u];
    }
#pragma unroll
    for (int u = 0; u < NBv; ++u) {
      const int e = tid + u * 256;
      sb[buf][e % BN][e / BN] = rb[u];
    }
  };
  load(0);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < Kd; k0 += BK) {
    const bool more = k0 + BK < Kd;
    if (more) load(k0 + BK);  // in flight during the MFMAs below
#pragma unroll
    for (int k = 0; k < BK; k += 2) {
      const int kk = k + (lane >> 5);
      float av[TA], bv[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) av[a] = sa[buf][wi + a * 32 + (lane & 31)][kk];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[b] = sb[buf][wj + b * 32 + (lane & 31)][kk];
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (more) {
      store(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gi = i0 + wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int gj = j0 + wj + b * 32 + (lane & 31);
        if (gi < M && gj < N) {
          float v = acc[a][b][r];
          if (den) v = v / den[gi];
          if (bias) v = v + bias[gj];
          out[(int64_t)gi * N + gj] = v;
        }
      }
}

// functional.py:54-67 maxpool (zero shadow row) / :6-22 nearest_upsample
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x, int N, int C,
                                                   const int64_t* __restrict__ nbr, int M, int H, int mode,
                                                   float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * C) return;
  const int m = (int)(e / C), c = (int)(e % C);
  if (mode == 1) {  // nearest upsample: first neighbour only
    const int64_t idx = nbr[(int64_t)m * H];
    out[e] = (idx >= N || idx < 0) ? 0.f : x[idx * C + c];
    return;
  }
  float best = -INFINITY;
  for (int h = 0; h < H; ++h) {
    const int64_t idx = nbr[(int64_t)m * H + h];
    const float v = (idx >= N || idx < 0) ? 0.f : x[idx * C + c];
    best = fmaxf(best, v);
  }
  out[e] = best;
}

// The same for C % 4 == 0 with four channels per thread and the neighbours four at a time: the indices of a step are loaded
// first (one broadcast address per row), then the four feature rows are in flight together -- the scalar kernel above walks
// H dependent (index -> row) round trips one after the other.
__global__ __launch_bounds__(256) void pool4_kernel(const float4* __restrict__ x, int N, int CV /* C / 4 */,
                                                    const int64_t* __restrict__ nbr, int M, int H, int mode,
                                                    float4* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)M * CV) return;
  const int m = (int)(e / CV), c = (int)(e % CV);
  const int64_t* row = nbr + (int64_t)m * H;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  if (mode == 1) {  // nearest upsample: first neighbour only
    const int64_t idx = row[0];
    out[e] = (idx >= N || idx < 0) ? zero : x[idx * CV + c];
    return;
  }
  float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int h0 = 0; h0 < H; h0 += 4) {
    int64_t idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) idx[u] = row[min(h0 + u, H - 1)];  // past the end: the last neighbour again (max is idempotent)
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool pad = idx[u] >= N || idx[u] < 0;
      v[u] = x[(pad ? 0 : idx[u]) * CV + c];  // unconditional load, the shadow row's zeros selected afterwards
      if (pad) v[u] = zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      best.x = fmaxf(best.x, v[u].x), best.y = fmaxf(best.y, v[u].y);
      best.z = fmaxf(best.z, v[u].z), best.w = fmaxf(best.w, v[u].w);
    }
  }
  out[e] = best;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_kpconv_workspace_bytes(int64_t n, int64_t m, int64_t k, int64_t cin) {
  if (n < 0 || m < 0 || k < 0 || cin < 0) return 0;
  return align_up((size_t)m * k * cin * sizeof(float), 256) + align_up((size_t)m * sizeof(float), 256) +
         align_up((size_t)n + 1, 256) + 1024;
}

extern "C" int gr_kpconv_forward(const float* s_feats, const float* q_points, const float* s_points,
                                 const int64_t* neighbor_indices, int64_t n, int64_t m, int64_t h, int64_t cin,
                                 int64_t cout, const float* kernel_points, int64_t k, const float* weights,
                                 const float* bias, float sigma, float inf, float* out, void* ws, size_t ws_bytes,
                                 void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && m >= 0 && h >= 0 && cin >= 1 && cout >= 1 && k >= 1, "bad sizes");
  GR_REQUIRE(k <= KP_MAX, "kernel_size must be <= %d", KP_MAX);
  GR_REQUIRE(m * k * cin < (1ll << 40) && n < (1ll << 31) && m < (1ll << 31), "sizes too large");
  if (m == 0) return GR_OK;
  GR_REQUIRE(s_feats && q_points && s_points && neighbor_indices && kernel_points && weights && out, "null argument");
  if (!ws || ws_bytes < gr_kpconv_workspace_bytes(n, m, k, cin)) {
    set_error("kpconv workspace too small");
    return GR_ERR_WORKSPACE;
  }
  char* p = static_cast<char*>(ws);
  float* WF = reinterpret_cast<float*>(p);
  p += align_up((size_t)m * k * cin * sizeof(float), 256);
  float* num = reinterpret_cast<float*>(p);
  p += align_up((size_t)m * sizeof(float), 256);
  uint8_t* flag = reinterpret_cast<uint8_t*>(p);
  KernelTimer timer("kpconv", stream);
  if (n > 0)
    hipLaunchKernelGGL(rowflag_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, s_feats, (int)n, (int)cin, flag);
  const size_t kp_lds = (size_t)std::min<int64_t>(h, KP_HMAX) * (KP_MAX + 2) * sizeof(float);
  const bool mfma = n > 0 && h <= 4 * KP_STEPS && (cin == 16 || cin == 32 || cin == 64 || cin == 128 || cin == 256);
  if (mfma) {
    const dim3 grid((unsigned)((m + 3) / 4)), blk(256);
#define GR_KP_MFMA(NT)                                                                                                       \
  hipLaunchKernelGGL((kp_gather_mfma_kernel<NT>), grid, blk, 0, stream, s_feats, q_points, s_points, neighbor_indices, (int)n, \
                     (int)m, (int)h, (int)k, kernel_points, sigma, inf, flag, WF, num)
    if (cin == 16) GR_KP_MFMA(1);
    else if (cin == 32) GR_KP_MFMA(2);
    else if (cin == 64) GR_KP_MFMA(4);
    else if (cin == 128) GR_KP_MFMA(8);
    else GR_KP_MFMA(16);
#undef GR_KP_MFMA
  } else if (cin <= 64)
    hipLaunchKernelGGL((kp_gather_kernel<64>), dim3((unsigned)m), dim3(64), kp_lds, stream, s_feats, q_points, s_points,
                       neighbor_indices, (int)n, (int)h, (int)cin, (int)k, kernel_points, sigma, inf, flag, WF, num);
  else if (cin <= 128)
    hipLaunchKernelGGL((kp_gather_kernel<128>), dim3((unsigned)m), dim3(128), kp_lds, stream, s_feats, q_points, s_points,
                       neighbor_indices, (int)n, (int)h, (int)cin, (int)k, kernel_points, sigma, inf, flag, WF, num);
  else
    hipLaunchKernelGGL((kp_gather_kernel<256>), dim3((unsigned)m), dim3(256), kp_lds, stream, s_feats, q_points, s_points,
                       neighbor_indices, (int)n, (int)h, (int)cin, (int)k, kernel_points, sigma, inf, flag, WF, num);
  const int kd = (int)(k * cin);
  const bool aligned = kd % BK == 0 && cout % 4 == 0 && cout >= 4 &&
                       ((reinterpret_cast<uintptr_t>(WF) | reinterpret_cast<uintptr_t>(weights)) & 15) == 0;
  if (m >= BT && cout > 64) {
    const int64_t blocks128 = ((cout + BT - 1) / BT) * ((m + BT - 1) / BT);
    if (blocks128 >= 768) {  // three or more 128-row workgroups per CU: the big tile's operand reuse wins
      const dim3 grid((unsigned)((cout + BT - 1) / BT), (unsigned)((m + BT - 1) / BT));
      if (aligned) hipLaunchKernelGGL((gemm_nn_big_kernel<128, 128, true>), grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
      else hipLaunchKernelGGL((gemm_nn_big_kernel<128, 128>), grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
    } else {
      const dim3 grid((unsigned)((cout + BT - 1) / BT), (unsigned)((m + 63) / 64));
      if (aligned) hipLaunchKernelGGL((gemm_nn_big_kernel<64, 128, true>), grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
      else hipLaunchKernelGGL((gemm_nn_big_kernel<64, 128>), grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
    }
  } else if (m >= BT && cout > 16) {
    // narrow outputs (the 32- and 64-channel stages): 128 x 64 tiles, same double-buffered pipeline
    const dim3 grid((unsigned)((cout + 63) / 64), (unsigned)((m + BT - 1) / BT));
    if (aligned) hipLaunchKernelGGL((gemm_nn_big_kernel<128, 64, true>), grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
    else hipLaunchKernelGGL((gemm_nn_big_kernel<128, 64>), grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
  } else {
    const dim3 grid((unsigned)((cout + GT - 1) / GT), (unsigned)((m + GT - 1) / GT));
    hipLaunchKernelGGL(gemm_nn_kernel, grid, dim3(256), 0, stream, WF, weights, (int)m, (int)cout, kd, num, bias, out);
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

namespace gr {
namespace {
// out[i, :] = data[index[i], :] for i < m (index_select along dim 0 of a 2-D fp32 tensor); rows move as float4 when the
// row length allows, one 16-lane group per row so a wave reads four 64-byte-aligned row pieces per instruction
template <typename VEC>
__device__ __forceinline__ VEC gather_poison();
template <>
__device__ __forceinline__ float gather_poison<float>() { return __uint_as_float(0x7fc00000u); }
template <>
__device__ __forceinline__ float4 gather_poison<float4>() {
  const float q = __uint_as_float(0x7fc00000u);
  return make_float4(q, q, q, q);
}

template <typename VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const VEC* __restrict__ data, int64_t n, int cv,
                                                          const int64_t* __restrict__ index, int64_t m,
                                                          VEC* __restrict__ out, int* __restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / cv;
  if (row >= m) return;
  const int col = (int)(t - row * cv);
  const int64_t src = index[row];
  if (src < 0 || src >= n) {
    if (col == 0) atomicOr(bad, 1);
    out[row * cv + col] = gather_poison<VEC>();  // never leave the row uninitialised: a quiet NaN marks it until the flag is read
    return;
  }
  out[row * cv + col] = data[src * cv + col];
}
}  // namespace
}  // namespace gr

extern "C" int gr_gather_rows(const float* data, int64_t n, int64_t c, const int64_t* index, int64_t m, float* out,
                              int* d_error_flag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && c >= 1 && m >= 0 && c < (1 << 24), "bad arguments");
  if (m == 0) return GR_OK;
  GR_REQUIRE(data && index && out && d_error_flag, "null argument");
  const bool v4 = (c % 4 == 0) && (reinterpret_cast<uintptr_t>(data) % 16 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  if (v4) {
    const int cv = (int)(c / 4);
    hipLaunchKernelGGL(gr::gather_rows_kernel<float4>, dim3((unsigned)((m * cv + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(data), n, cv, index, m, reinterpret_cast<float4*>(out), d_error_flag);
  } else {
    hipLaunchKernelGGL(gr::gather_rows_kernel<float>, dim3((unsigned)((m * c + 255) / 256)), dim3(256), 0, stream, data, n,
                       (int)c, index, m, out, d_error_flag);
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

extern "C" int gr_neighbor_pool(const float* x, int64_t n, int64_t c, const int64_t* neighbor_indices, int64_t m,
                                int64_t h, int mode, float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && c >= 1 && m >= 0 && h >= 1 && (mode == 0 || mode == 1), "bad arguments");
  if (m == 0) return GR_OK;
  GR_REQUIRE(x && neighbor_indices && out, "null argument");
  if (c % 4 == 0 && n > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(pool4_kernel, dim3((unsigned)((m * (c / 4) + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x), (int)n, (int)(c / 4), neighbor_indices, (int)m, (int)h, mode,
                       reinterpret_cast<float4*>(out));
  else
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((m * c + 255) / 256)), dim3(256), 0, stream, x, (int)n, (int)c,
                       neighbor_indices, (int)m, (int)h, mode, out);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

// ---------------------------------------------------------------- GroupNorm over a (N, C) point-feature matrix
// geotransformer/modules/kpconv/modules.py:32-50 transposes the matrix to (1, C, N) and calls nn.GroupNorm: one mean and
// one variance per group of C / G channels over ALL N points.  ATen's kernel gives that shape one workgroup per group (32
// rows of 60 000 x C/32 values: 5.6 ms of the 16 ms backbone, plus 2.3 ms for the two transposes); here the matrix stays
// (N, C): pass 1 streams it once with float4 loads and leaves per-workgroup (sum, sum of squares) partials in fp64, one
// workgroup folds them into mean / rstd per group, pass 2 streams the matrix again and applies (x - mean) * rstd * gamma + beta (+ the LeakyReLU that always follows, modules.py:75,138).
namespace gr {
namespace {

constexpr int GN_T = 256;
constexpr int GN_BLOCKS = 256;  // pass 1 workgroups (grid-stride over row tiles)
constexpr int GN_MAXG = 64;

// The float4 a thread reads belongs to columns 4*col4 .. 4*col4+3, fixed for the thread when C/4 <= 256 (the row index
// advances instead); wider rows give every thread C/1024 column positions.
// seg_off (optional, device, gridDim.y + 1 entries): rows [seg_off[s], seg_off[s+1]) are normalised on their own -- several
// stack-mode batches (scene pairs) in one launch, each with the statistics it would have had alone.
template <bool APPLY>
__global__ __launch_bounds__(GN_T) void group_norm_kernel(const float* __restrict__ x, int64_t n, int c, int groups,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float slope, double* __restrict__ partial, int nblk_a,
                                                          float* __restrict__ out, const int64_t* __restrict__ seg_off,
                                                          const float* __restrict__ residual) {
  __shared__ double s_acc[2 * GN_MAXG];
  if (seg_off != nullptr) {
    const int64_t r0 = seg_off[blockIdx.y];
    n = seg_off[blockIdx.y + 1] - r0;
    x += r0 * c;
    out += r0 * c;
    if (residual) residual += r0 * c;
    partial += (int64_t)blockIdx.y * ((int64_t)nblk_a * 2 * groups + groups);  // partials, then 2 G floats (= G doubles) of stats
  }
  __shared__ float s_mean[GN_MAXG], s_rstd[GN_MAXG];
  const int cols4 = c / 4, cg = c / groups;
  const int tid = threadIdx.x;
  const int npos = cols4 > GN_T ? cols4 / GN_T : 1;          // column positions per thread
  const int rows_per_pass = cols4 >= GN_T ? 1 : GN_T / cols4;  // rows a workgroup covers per step
  const int col_base = cols4 >= GN_T ? tid : tid % cols4;
  const int rsub = cols4 >= GN_T ? 0 : tid / cols4;
  if (APPLY) {
    // mean / rstd of every group, left behind the partials by group_norm_stats_kernel
    const float* stats = reinterpret_cast<const float*>(partial + (int64_t)nblk_a * 2 * groups);
    if (tid < groups) {
      s_mean[tid] = stats[tid];
      s_rstd[tid] = stats[groups + tid];
    }
    __syncthreads();
  } else {
    for (int i = tid; i < 2 * groups; i += GN_T) s_acc[i] = 0.0;
    __syncthreads();
  }
  for (int p = 0; p < npos; ++p) {
    const int col4 = col_base + p * GN_T;
    const int ch = 4 * col4;
    float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
    float mean[4], rstd[4];
    if (APPLY) {
      if (gamma) ga = *reinterpret_cast<const float4*>(gamma + ch);
      if (beta) be = *reinterpret_cast<const float4*>(beta + ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mean[j] = s_mean[(ch + j) / cg];
        rstd[j] = s_rstd[(ch + j) / cg];
      }
    }
    double sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    for (int64_t r = (int64_t)blockIdx.x * rows_per_pass + rsub; r < n; r += (int64_t)gridDim.x * rows_per_pass) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * c + ch);
      if (APPLY) {
        float4 y;
        y.x = (v.x - mean[0]) * rstd[0] * ga.x + be.x;
        y.y = (v.y - mean[1]) * rstd[1] * ga.y + be.y;
        y.z = (v.z - mean[2]) * rstd[2] * ga.z + be.z;
        y.w = (v.w - mean[3]) * rstd[3] * ga.w + be.w;
        if (residual) {  // the residual block's shortcut, added before the activation (modules.py:138: leaky_relu(x + shortcut))
          const float4 sc = *reinterpret_cast<const float4*>(residual + r * c + ch);
          y.x += sc.x, y.y += sc.y, y.z += sc.z, y.w += sc.w;
        }
        y.x = y.x >= 0.f ? y.x : y.x * slope;
        y.y = y.y >= 0.f ? y.y : y.y * slope;
        y.z = y.z >= 0.f ? y.z : y.z * slope;
        y.w = y.w >= 0.f ? y.w : y.w * slope;
        *reinterpret_cast<float4*>(out + r * c + ch) = y;
      } else {
        sum[0] += v.x, sum[1] += v.y, sum[2] += v.z, sum[3] += v.w;
        sq[0] += (double)v.x * v.x, sq[1] += (double)v.y * v.y, sq[2] += (double)v.z * v.z, sq[3] += (double)v.w * v.w;
      }
    }
    if (!APPLY) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int g = (ch + j) / cg;
        atomicAdd(&s_acc[g], sum[j]);
        atomicAdd(&s_acc[groups + g], sq[j]);
      }
    }
  }
  if (!APPLY) {
    __syncthreads();
    for (int i = tid; i < 2 * groups; i += GN_T) partial[(int64_t)blockIdx.x * 2 * groups + i] = s_acc[i];
  }
}

// one workgroup: fold the pass-1 partials (nblk x 2G doubles) into mean / rstd per group, four lanes per (group, moment)
__global__ __launch_bounds__(GN_T) void group_norm_stats_kernel(double* __restrict__ partial, int nblk, int groups, int64_t n,
                                                                int cg, float eps, const int64_t* __restrict__ seg_off) {
  __shared__ double s_tot[2 * GN_MAXG];
  if (seg_off != nullptr) {
    n = seg_off[blockIdx.x + 1] - seg_off[blockIdx.x];
    partial += (int64_t)blockIdx.x * ((int64_t)nblk * 2 * groups + groups);
  }
  const int tid = threadIdx.x, item = tid / 4, sub = tid % 4;  // 2 * groups <= 128 items x 4 lanes <= 512: loop below
  for (int it = item; it < 2 * groups; it += GN_T / 4) {
    double acc = 0.0;
    for (int b = sub; b < nblk; b += 4) acc += partial[(int64_t)b * 2 * groups + it];
    acc += __shfl_xor(acc, 1, WAVE);
    acc += __shfl_xor(acc, 2, WAVE);
    if (sub == 0) s_tot[it] = acc;
  }
  __syncthreads();
  float* stats = reinterpret_cast<float*>(partial + (int64_t)nblk * 2 * groups);
  if (tid < groups) {
    const double cnt = fmax((double)n * (double)cg, 1.0);
    const double mean = s_tot[tid] / cnt;
    const double var = fmax(s_tot[groups + tid] / cnt - mean * mean, 0.0);
    stats[tid] = (float)mean;
    stats[groups + tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

}  // namespace
}  // namespace gr

extern "C" size_t gr_group_norm_workspace_bytes(int64_t groups) {
  return groups > 0 ? ((size_t)gr::GN_BLOCKS * 2 * (size_t)groups + (size_t)groups) * sizeof(double) + 512 : 0;
}

static int group_norm_impl(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta,
                           float eps, float negative_slope, float* out, const int64_t* seg_off, int64_t nseg, int64_t max_seg_rows,
                           void* ws, size_t ws_bytes, hipStream_t stream, const float* residual = nullptr) {
  GR_REQUIRE(n >= 0 && c >= 4 && groups >= 1 && groups <= gr::GN_MAXG && c % groups == 0, "group_norm: bad sizes");
  const int64_t cols4 = c / 4;
  GR_REQUIRE(c % 4 == 0 && ((cols4 <= gr::GN_T && gr::GN_T % cols4 == 0) || (cols4 > gr::GN_T && cols4 % gr::GN_T == 0)),
             "group_norm: %lld channels are not supported (C / 4 must divide 256 or be a multiple of it)", (long long)c);
  if (n == 0 || nseg == 0) return GR_OK;
  GR_REQUIRE(x && out, "null argument");
  GR_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0, "group_norm: unaligned tensors");
  const bool seg = seg_off != nullptr;
  const int64_t rows = seg ? max_seg_rows : n;  // rows of the longest segment: sizes the grid
  const int rows_per_pass = cols4 >= gr::GN_T ? 1 : (int)(gr::GN_T / cols4);
  const int tiles = (int)std::min<int64_t>(std::max<int64_t>((rows + rows_per_pass - 1) / rows_per_pass, 1), 1 << 20);
  // pass 1 (read-only) wants the whole device streaming: ~2 048 workgroups over all segments, at most GN_BLOCKS partials each
  const int cap_a = seg ? std::min(gr::GN_BLOCKS, std::max(1, 2048 / (int)nseg)) : gr::GN_BLOCKS;
  const int nblk_a = std::min(cap_a, tiles);
  const int nblk_b = std::min(tiles, seg ? std::max(8, 4096 / (int)std::min<int64_t>(nseg, 512)) : 4096);
  const size_t per_seg = ((size_t)nblk_a * 2 * groups + groups) * sizeof(double);
  if (!ws || ws_bytes < per_seg * (size_t)(seg ? nseg : 1) + 256) {
    gr::set_error("group_norm workspace too small");
    return GR_ERR_WORKSPACE;
  }
  double* partial = static_cast<double*>(ws);
  const unsigned gy = seg ? (unsigned)nseg : 1u;
  gr::KernelTimer timer("group_norm", stream);
  hipLaunchKernelGGL(gr::group_norm_kernel<false>, dim3(nblk_a, gy), dim3(gr::GN_T), 0, stream, x, n, (int)c, (int)groups, gamma,
                     beta, eps, negative_slope, partial, nblk_a, out, seg_off, (const float*)nullptr);
  hipLaunchKernelGGL(gr::group_norm_stats_kernel, dim3(gy), dim3(gr::GN_T), 0, stream, partial, nblk_a, (int)groups, n,
                     (int)(c / groups), eps, seg_off);
  GR_REQUIRE(residual == nullptr || reinterpret_cast<uintptr_t>(residual) % 16 == 0, "group_norm: unaligned residual");
  hipLaunchKernelGGL(gr::group_norm_kernel<true>, dim3(nblk_b, gy), dim3(gr::GN_T), 0, stream, x, n, (int)c, (int)groups, gamma,
                     beta, eps, negative_slope, partial, nblk_a, out, seg_off, residual);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

extern "C" int gr_group_norm(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta,
                             float eps, float negative_slope, float* out, void* ws, size_t ws_bytes, void* stream_) {
  return group_norm_impl(x, n, c, groups, gamma, beta, eps, negative_slope, out, nullptr, 1, n, ws, ws_bytes,
                         static_cast<hipStream_t>(stream_));
}

extern "C" size_t gr_group_norm_seg_workspace_bytes(int64_t groups, int64_t nseg) {
  return groups > 0 && nseg > 0 ? (size_t)nseg * ((size_t)gr::GN_BLOCKS * 2 * groups + groups) * sizeof(double) + 256 : 0;
}

extern "C" int gr_group_norm_seg(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta,
                                 float eps, float negative_slope, float* out, const int64_t* seg_off, int64_t nseg,
                                 int64_t max_seg_rows, void* ws, size_t ws_bytes, void* stream_) {
  GR_REQUIRE(seg_off != nullptr && nseg >= 0 && nseg < 65536 && max_seg_rows >= 0, "group_norm_seg: bad segments");
  return group_norm_impl(x, n, c, groups, gamma, beta, eps, negative_slope, out, seg_off, nseg, max_seg_rows, ws, ws_bytes,
                         static_cast<hipStream_t>(stream_));
}

// GroupNorm -> + residual -> LeakyReLU in the apply pass: the tail of a residual block (modules.py:135-138: unary2's norm
// has no activation, then leaky_relu(x + shortcut)) without the two extra passes over the (N, C) matrix.  seg_off null: one
// segment (nseg and max_seg_rows ignored); residual null: plain gr_group_norm(_seg).
extern "C" int gr_group_norm_res(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta,
                                 float eps, float negative_slope, const float* residual, float* out, const int64_t* seg_off,
                                 int64_t nseg, int64_t max_seg_rows, void* ws, size_t ws_bytes, void* stream_) {
  if (seg_off == nullptr)
    return group_norm_impl(x, n, c, groups, gamma, beta, eps, negative_slope, out, nullptr, 1, n, ws, ws_bytes,
                           static_cast<hipStream_t>(stream_), residual);
  GR_REQUIRE(nseg >= 0 && nseg < 65536 && max_seg_rows >= 0, "group_norm_res: bad segments");
  return group_norm_impl(x, n, c, groups, gamma, beta, eps, negative_slope, out, seg_off, nseg, max_seg_rows, ws, ws_bytes,
                         static_cast<hipStream_t>(stream_), residual);
}
