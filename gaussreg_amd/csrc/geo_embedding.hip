// GeometricStructureEmbedding forward ("next" row):
//   geotransformer/modules/geotransformer/geotransformer.py:26-73   (indices + the two projections)
//   geotransformer/modules/transformer/positional_embedding.py:8-34 (sinusoidal embedding)
// The reference materialises, per cloud of N superpoints, d_indices (N,N), a_indices (N,N,k), two sinusoidal
// embeddings (N,N,C) and (N,N,k,C), and runs 1+k Linear(C,C) layers over them: 2*(1+k)*N^2*C^2 flop
// (308 GFLOP at N=767, C=256, k=3) and ~1.8 GB of temporaries.  Here one kernel per cloud produces the final
// (N,N,C) tensor directly:
//   * a workgroup owns 128 consecutive (n,m) pairs x 256 output channels;
//   * per pair it computes the distance index and the k angular indices once (LDS),
//   * the sinusoidal rows are generated on the fly, 16 K-values at a time, straight into the A tile of an
//     fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32) whose B tile streams W_a / W_d out of L2, double-buffered;
//   * the k angular GEMMs are reduced (max / mean) in registers, the distance GEMM is added, biases applied,
//     and only the result is written: the output (N*N*C*4 B) is the only HBM traffic that scales with N^2.
// Float op order differs from ATen (GEMM summation order), so parity is a tolerance (tests/test_gpu_next.py).
#include <type_traits>

#include "common.hpp"

namespace gr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GE_ROWS = 128;  // (n,m) pairs per workgroup
constexpr int GE_COLS = 256;  // output channels per workgroup
constexpr int GE_K = 32;      // K slab
constexpr int GE_LD = GE_K + 1;
constexpr int GE_T = 512;
constexpr int GE_KMAX = 8;    // angle_k <= 8
constexpr int GE_SPLIT_CMAX = 512;  // hidden_dim up to which the split-bf16 kernel's weight planes fit the workspace

__device__ __forceinline__ float sq_dist_ref(const float3 a, float a2, const float3 b, float b2) {
  // pairwise_distance.py:21-31: xy by matmul, then (x2 - 2 xy) + y2, clamped at 0
  const float xy = fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
  return fmaxf((a2 - 2.0f * xy) + b2, 0.0f);
}
__device__ __forceinline__ float3 ld3(const float* p, int i) { return make_float3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ float norm2(const float3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }


// k nearest other points per point: geotransformer.py:42 topk(k+1, largest=False)[1][:, :, 1:]
// (ascending distance, the first -- the point itself -- dropped; ties: lowest index first).  One wave per point.
__global__ __launch_bounds__(256) void geo_knn_kernel(const float* __restrict__ pts, int n, int k,
                                                      int32_t* __restrict__ knn) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float3 p = ld3(pts, row);
  const float p2 = norm2(p);
  unsigned long long best[GE_KMAX + 1];
#pragma unroll
  for (int i = 0; i <= GE_KMAX; ++i) best[i] = ~0ull;
  for (int m = lane; m < n; m += 64) {
    const float3 q = ld3(pts, m);
    const float d = sqrtf(sq_dist_ref(p, p2, q, norm2(q)));
    unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)m;
#pragma unroll
    for (int i = 0; i <= GE_KMAX; ++i) {  // sorted insertion (register-resident)
      const unsigned long long lo = key < best[i] ? key : best[i];
      key = key < best[i] ? best[i] : key;
      best[i] = lo;
    }
  }
  for (int s = 0; s <= k; ++s) {
    unsigned long long v = best[0];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long o = __shfl_xor(v, d, 64);
      v = o < v ? o : v;
    }
    if (best[0] == v) {  // the owner pops (keys are unique: they carry the index)
#pragma unroll
      for (int i = 0; i < GE_KMAX; ++i) best[i] = best[i + 1];
      best[GE_KMAX] = ~0ull;
    }
    if (s > 0 && lane == 0) knn[row * k + (s - 1)] = v == ~0ull ? row : (int)(unsigned)(v & 0xffffffffull);
  }
}

// per-pair embedding indices (geotransformer.py:38-55) for the GE_ROWS consecutive (a, b) pairs of a workgroup:
// xs[0..k-1][row] = angular indices, xs[k][row] = distance index
__device__ __forceinline__ void ge_pair_indices(const float* __restrict__ pts, int n, const int32_t* __restrict__ knn,
                                                int k, float sigma_d, float factor_a, int64_t r0, int64_t total, int tid,
                                                float (*xs)[GE_ROWS]) {
  if (tid < GE_ROWS) {
    const int64_t r = r0 + tid;
    float xd = 0.f, xa[GE_KMAX];
#pragma unroll
    for (int i = 0; i < GE_KMAX; ++i) xa[i] = 0.f;
    if (r < total) {
      const int a = (int)(r / n), b = (int)(r - (int64_t)a * n);
      const float3 pa = ld3(pts, a), pb = ld3(pts, b);
      xd = sqrtf(sq_dist_ref(pa, norm2(pa), pb, norm2(pb))) / sigma_d;
      const float3 anc = make_float3(pb.x - pa.x, pb.y - pa.y, pb.z - pa.z);
#pragma unroll
      for (int i = 0; i < GE_KMAX; ++i)
        if (i < k) {
          const float3 pk = ld3(pts, knn[a * k + i]);
          const float3 ref = make_float3(pk.x - pa.x, pk.y - pa.y, pk.z - pa.z);
          const float3 cr = make_float3(ref.y * anc.z - ref.z * anc.y, ref.z * anc.x - ref.x * anc.z,
                                        ref.x * anc.y - ref.y * anc.x);
          const float sn = sqrtf(norm2(cr));
          // torch.sum accumulates from +0: an all-(-0) product row (a == b, anc = +0) must give +0, not -0 (atan2 -> pi)
          const float cs = ((0.0f + ref.x * anc.x) + ref.y * anc.y) + ref.z * anc.z;
          xa[i] = atan2f(sn, cs) * factor_a;
        }
    }
#pragma unroll
    for (int i = 0; i < GE_KMAX; ++i)
      if (i < k) xs[i][tid] = xa[i];
    xs[k][tid] = xd;
  }
}

__global__ __launch_bounds__(GE_T) void geo_embedding_kernel(
    const float* __restrict__ pts, int n, const int32_t* __restrict__ knn, int k, const float* __restrict__ w_d,
    const float* __restrict__ b_d, const float* __restrict__ w_a, const float* __restrict__ b_a,
    const float* __restrict__ div_term, int C, float sigma_d, float factor_a, int mean, float* __restrict__ out) {
  __shared__ float sa[2][GE_ROWS][GE_LD];
  __shared__ float sb[2][GE_COLS][GE_LD];
  __shared__ float xs[GE_KMAX + 1][GE_ROWS];  // [0..k-1] angular indices, [k] distance index
  const int64_t total = (int64_t)n * n;
  const int64_t r0 = (int64_t)blockIdx.x * GE_ROWS;
  const int j0 = blockIdx.y * GE_COLS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 2) * 64, wj = (w & 3) * 64;

  ge_pair_indices(pts, n, knn, k, sigma_d, factor_a, r0, total, tid, xs);
  __syncthreads();

  f32x16 acc[2][2], red[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f, red[a][b][r] = 0.f;

  // staging registers: A = 128 rows x 8 frequencies (sin, cos) -> 2 per thread; B = 256 cols x 16 k -> 8 per thread
  constexpr int NA = GE_ROWS * (GE_K / 2) / GE_T, NB = GE_COLS * GE_K / GE_T, NF = GE_K / 2;
  float ra_s[NA], ra_c[NA], rb[NB];
  const int slabs = C / GE_K;  // C % 16 == 0 (checked by the host)
  auto gen = [&](int phase, int k0) {
    const float* W = phase < k ? w_a : w_d;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + u * GE_T;
      const int r = e / NF, f = e % NF;
      const float omega = xs[phase][r] * div_term[(k0 >> 1) + f];  // positional_embedding.py:27
      sincosf(omega, &ra_s[u], &ra_c[u]);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = tid + u * GE_T;
      const int j = e / GE_K, kk = e % GE_K;
      const int gj = j0 + j;
      rb[u] = gj < C ? W[(int64_t)gj * C + k0 + kk] : 0.f;  // nn.Linear: y = x W^T
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + u * GE_T;
      const int r = e / NF, f = e % NF;
      sa[buf][r][2 * f] = ra_s[u];  // positional_embedding.py:30-31: (sin, cos) interleaved
      sa[buf][r][2 * f + 1] = ra_c[u];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = tid + u * GE_T;
      sb[buf][e / GE_K][e % GE_K] = rb[u];
    }
  };

  const int steps = (k + 1) * slabs;
  gen(0, 0);
  store(0);
  __syncthreads();
  int buf = 0;
  for (int s = 0; s < steps; ++s) {
    const int phase = s / slabs;
    const bool more = s + 1 < steps;
    if (more) gen((s + 1) / slabs, ((s + 1) % slabs) * GE_K);  // overlaps the MFMAs below
#pragma unroll
    for (int kk2 = 0; kk2 < GE_K; kk2 += 2) {
      const int kk = kk2 + (lane >> 5);
      const float a0 = sa[buf][wi + (lane & 31)][kk], a1 = sa[buf][wi + 32 + (lane & 31)][kk];
      const float b0 = sb[buf][wj + (lane & 31)][kk], b1 = sb[buf][wj + 32 + (lane & 31)][kk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if ((s + 1) % slabs == 0 && phase < k) {
      // one angular projection finished: fold it into the reduction over the k neighbours (geotransformer.py:65-68)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[a][b][r];
            red[a][b][r] = phase == 0 ? v : (mean ? red[a][b][r] + v : fmaxf(red[a][b][r], v));
            acc[a][b][r] = 0.f;
          }
    }
    if (more) {
      store(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  // ---- epilogue: (proj_d + b_d) + reduce_k(proj_a + b_a)
  const float inv_k = k > 0 ? 1.0f / (float)k : 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int gj = j0 + wj + b * 32 + (lane & 31);
      if (gj >= C) continue;
      const float bd = b_d[gj], ba = k > 0 ? b_a[gj] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gi = r0 + wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gi < total) {
          const float av = k > 0 ? (mean ? (red[a][b][r] + (float)k * ba) * inv_k : red[a][b][r] + ba) : 0.f;
          out[gi * C + gj] = (acc[a][b][r] + bd) + av;
        }
      }
    }
}


// ---------------------------------------------------------------- split-bf16 variant of the same GEMMs
// The bf16 matrix pipe of gfx950 is 16x the fp32 one.  Every fp32 operand is split exactly into three bf16 parts,
// x = hi + mid + lo (8 mantissa bits each), and a product a*b is evaluated as the six part products of weight
// >= 2^-18 relative: hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  What is dropped (mid*lo, lo*mid, lo*lo) is below 2^-26 of the product -- smaller than
// one fp32 rounding -- so the result differs from the fp32-MFMA kernel only by summation order, at 6/16 of its
// matrix-pipe time.  Fragment layout (probed on the hardware): A lane l = row l & 31, k = 8 * (l >> 5) + 0..7.

// sin and cos for moderate arguments (|x| < 2^11; the embedding's index * div_term is a few tens): three-constant
// Cody-Waite reduction by pi/2 and the Cephes single-precision polynomials on [-pi/4, pi/4] (~1 ulp).  Used by the
// split-bf16 kernel, where the operand generation competes with a 2.7x faster matrix pipe; workgroups that see a larger
// index (or a non-finite one) use the library sincosf().
__device__ __forceinline__ void sincos_moderate(float x, float* sn, float* cs) {
  const float kf = rintf(x * 0.636619772367581343f);  // 2/pi
  float r = fmaf(kf, -1.5703125f, x);                   // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
  r = fmaf(kf, -4.837512969970703125e-4f, r);
  r = fmaf(kf, -7.54978995489188e-8f, r);
  const float r2 = r * r;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, r2, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  const float s = fmaf(ps * r2, r, r);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, r2, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  const float c = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)kf;
  const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int GS_K = 16;            // K slab = one MFMA k-step
constexpr int GS_LD = GS_K + 8;     // bf16 per LDS row (48 B: ds_read_b128 of 16 rows hits 16 disjoint bank quads)

// x = hi + mid + lo with three bf16 parts taken by truncation: every subtraction is exact and the three parts carry
// the 24 significant bits of x (what is left is below one ulp of x).  Returned in the HIGH halves of the words.
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mid);
  lo = __float_as_uint(r2) & 0xffff0000u;
}

// W (c x c fp32, row = output channel) -> three bf16 planes of the same shape
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, int64_t n,
                                                            unsigned short* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h, m, l;
  split3(w[i], h, m, l);
  out[i] = (unsigned short)(h >> 16);
  out[n + i] = (unsigned short)(m >> 16);
  out[2 * n + i] = (unsigned short)(l >> 16);
}

__global__ __launch_bounds__(GE_T) void geo_embedding_split_kernel(
    const float* __restrict__ pts, int n, const int32_t* __restrict__ knn, int k,
    const unsigned short* __restrict__ wd3, const float* __restrict__ b_d, const unsigned short* __restrict__ wa3,
    const float* __restrict__ b_a, const float* __restrict__ div_term, int C, float sigma_d, float factor_a, int mean,
    float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned short sa[2][3][GE_ROWS][GS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short sb[2][3][GE_COLS][GS_LD];
  __shared__ float xs[GE_KMAX + 1][GE_ROWS];
  const int64_t total = (int64_t)n * n;
  const int64_t r0 = (int64_t)blockIdx.x * GE_ROWS;
  const int j0 = blockIdx.y * GE_COLS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 2) * 64, wj = (w & 3) * 64;
  __shared__ int s_big;
  if (tid == 0) s_big = 0;
  __syncthreads();
  ge_pair_indices(pts, n, knn, k, sigma_d, factor_a, r0, total, tid, xs);
  if (tid < GE_ROWS) {
    float mx = 0.f;
    for (int i = 0; i <= k; ++i) mx = fmaxf(mx, fabsf(xs[i][tid]));
    if (!(mx < 2048.0f) || isnan(mx)) s_big = 1;
    for (int i = 0; i <= k; ++i)
      if (isnan(xs[i][tid])) s_big = 1;
  }
  __syncthreads();
  const bool big = s_big != 0;

  f32x16 acc[2][2], red[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f, red[a][b][r] = 0.f;

  // staging: A = 128 rows x 8 frequencies -> 2 (row, f) per thread, each (sin, cos) x 3 parts packed in three words;
  //          B = 3 planes x 256 cols x 16 k bf16 = 1536 16-byte chunks -> 3 per thread
  constexpr int NA = GE_ROWS * (GS_K / 2) / GE_T, NF = GS_K / 2, NBC = 3 * GE_COLS * 2 / GE_T;
  unsigned ra[NA][3];
  uint4 rbv[1][NBC];
  const int slabs = C / GS_K;
  const int64_t plane = (int64_t)C * C;
  auto gen_a = [&](int phase, int k0) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + u * GE_T;
      const int r = e / NF, f = e % NF;
      const float omega = xs[phase][r] * div_term[(k0 >> 1) + f];  // positional_embedding.py:27
      float sn, cs;
      if (big) sincosf(omega, &sn, &cs);
      else sincos_moderate(omega, &sn, &cs);
      unsigned sh, sm, sl, ch, cm, cl;
      split3(sn, sh, sm, sl);
      split3(cs, ch, cm, cl);
      ra[u][0] = (sh >> 16) | ch;  // positional_embedding.py:30-31: (sin, cos) interleaved along k
      ra[u][1] = (sm >> 16) | cm;
      ra[u][2] = (sl >> 16) | cl;
    }
  };
  auto load_b = [&](auto slot, int phase, int k0) {
    constexpr int R = decltype(slot)::value;
    const unsigned short* W3 = phase < k ? wa3 : wd3;
#pragma unroll
    for (int u = 0; u < NBC; ++u) {
      const int e = tid + u * GE_T;
      const int p = e / (GE_COLS * 2), rem = e % (GE_COLS * 2);
      const int col = rem >> 1, half = rem & 1;
      const int gj = min(j0 + col, C - 1);
      rbv[R][u] = *reinterpret_cast<const uint4*>(W3 + p * plane + (int64_t)gj * C + k0 + half * 8);
    }
  };
  auto store = [&](auto slot, int buf) {
    constexpr int R = decltype(slot)::value;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + u * GE_T;
      const int r = e / NF, f = e % NF;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(&sa[buf][p][r][2 * f]) = ra[u][p];
    }
#pragma unroll
    for (int u = 0; u < NBC; ++u) {
      const int e = tid + u * GE_T;
      const int p = e / (GE_COLS * 2), rem = e % (GE_COLS * 2);
      *reinterpret_cast<uint4*>(&sb[buf][p][rem >> 1][(rem & 1) * 8]) = rbv[R][u];
    }
  };

  const int steps = (k + 1) * slabs;
  using Slot0 = std::integral_constant<int, 0>;
  load_b(Slot0{}, 0, 0);
  gen_a(0, 0);
  store(Slot0{}, 0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    const int phase = s / slabs;
    const bool more = s + 1 < steps;
    if (more) {  // next slab: weights requested, operands generated while this slab's MFMAs run
      load_b(Slot0{}, (s + 1) / slabs, ((s + 1) % slabs) * GS_K);
      gen_a((s + 1) / slabs, ((s + 1) % slabs) * GS_K);
    }
    {
      const int ra_ = wi + (lane & 31), rb_ = wj + (lane & 31), kh = (lane >> 5) * 8;
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[0][p] = *reinterpret_cast<const bf16x8*>(&sa[buf][p][ra_][kh]);
        fa[1][p] = *reinterpret_cast<const bf16x8*>(&sa[buf][p][ra_ + 32][kh]);
        fb[0][p] = *reinterpret_cast<const bf16x8*>(&sb[buf][p][rb_][kh]);
        fb[1][p] = *reinterpret_cast<const bf16x8*>(&sb[buf][p][rb_ + 32][kh]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          f32x16 c = acc[a][b];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], c, 0, 0, 0);  // lo  * hi
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], c, 0, 0, 0);  // hi  * lo
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], c, 0, 0, 0);  // mid * mid
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], c, 0, 0, 0);  // mid * hi
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], c, 0, 0, 0);  // hi  * mid
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], c, 0, 0, 0);  // hi  * hi
          acc[a][b] = c;
        }
    }
    if ((s + 1) % slabs == 0 && phase < k) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[a][b][r];
            red[a][b][r] = phase == 0 ? v : (mean ? red[a][b][r] + v : fmaxf(red[a][b][r], v));
            acc[a][b][r] = 0.f;
          }
    }
    if (more) {
      store(Slot0{}, buf ^ 1);
      __syncthreads();
    }
  }
  const float inv_k = k > 0 ? 1.0f / (float)k : 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int gj = j0 + wj + b * 32 + (lane & 31);
      if (gj >= C) continue;
      const float bd = b_d[gj], ba = k > 0 ? b_a[gj] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gi = r0 + wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gi < total) {
          const float av = k > 0 ? (mean ? (red[a][b][r] + (float)k * ba) * inv_k : red[a][b][r] + ba) : 0.f;
          out[gi * C + gj] = (acc[a][b][r] + bd) + av;
        }
      }
    }
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_geo_embedding_workspace_bytes(int64_t n, int64_t angle_k) {
  if (n < 0 || angle_k < 0) return 0;
  return align_up((size_t)n * (size_t)std::max<int64_t>(angle_k, 1) * 4, 256) + 256 +
         2 * align_up((size_t)3 * GE_SPLIT_CMAX * GE_SPLIT_CMAX * 2, 256);  // split weights (bf16 x 3) of proj_d, proj_a
}

extern "C" int gr_geo_embedding(const float* points, int64_t n, const float* w_d, const float* b_d, const float* w_a,
                                const float* b_a, const float* div_term, int64_t c, float sigma_d, float factor_a,
                                int64_t angle_k, int reduction_mean, float* out, void* ws, size_t ws_bytes,
                                void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && n < 46341, "geo_embedding: n*n must fit int32 pair ids per row (n=%lld)", (long long)n);
  GR_REQUIRE(c > 0 && c % 16 == 0, "geo_embedding: hidden_dim must be a positive multiple of 16 (got %lld)", (long long)c);
  GR_REQUIRE(angle_k >= 0 && angle_k <= GE_KMAX, "geo_embedding: angle_k must be in [0, %d]", GE_KMAX);
  GR_REQUIRE(angle_k < n || n == 0, "geo_embedding: angle_k (%lld) needs more than %lld points", (long long)angle_k, (long long)n);
  if (n == 0) return GR_OK;
  GR_REQUIRE(points && w_d && b_d && div_term && out && (angle_k == 0 || (w_a && b_a)), "null argument");
  if (!ws || ws_bytes < gr_geo_embedding_workspace_bytes(n, angle_k)) {
    set_error("geo_embedding workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver cv(ws);
  int32_t* knn = cv.take<int32_t>((size_t)n * std::max<int64_t>(angle_k, 1));
  if (angle_k > 0)
    hipLaunchKernelGGL(geo_knn_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, points, (int)n, (int)angle_k, knn);
  const bool fp32_mfma = (reduction_mean & 2) != 0 || c > GE_SPLIT_CMAX || c % 16 != 0;
  const int mean = reduction_mean & 1;
  const dim3 grid((unsigned)((n * n + GE_ROWS - 1) / GE_ROWS), (unsigned)((c + GE_COLS - 1) / GE_COLS));
  if (fp32_mfma) {
    GR_REQUIRE(c % GE_K == 0, "geo_embedding: the fp32-MFMA kernel needs hidden_dim %% 32 == 0 (got %lld)", (long long)c);
    KernelTimer timer("geo_embedding", stream);
    hipLaunchKernelGGL(geo_embedding_kernel, grid, dim3(GE_T), 0, stream, points, (int)n, knn, (int)angle_k, w_d, b_d,
                       w_a, b_a, div_term, (int)c, sigma_d, factor_a, mean, out);
  } else {
    unsigned short* wd3 = cv.take<unsigned short>((size_t)3 * GE_SPLIT_CMAX * GE_SPLIT_CMAX);
    unsigned short* wa3 = cv.take<unsigned short>((size_t)3 * GE_SPLIT_CMAX * GE_SPLIT_CMAX);
    const int64_t nw = c * c;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, w_d, nw, wd3);
    if (angle_k > 0)
      hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, w_a, nw, wa3);
    KernelTimer timer("geo_embedding", stream);
    hipLaunchKernelGGL(geo_embedding_split_kernel, grid, dim3(GE_T), 0, stream, points, (int)n, knn, (int)angle_k, wd3,
                       b_d, wa3, b_a, div_term, (int)c, sigma_d, factor_a, mean, out);
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

namespace gr {
namespace {

// ---------------------------------------------------------------- the same embedding from two function tables
// Both projections act on a sinusoidal embedding of ONE scalar: out[a, b, :] = F_d(d_ab / sigma_d) + red_i F_a(theta_abi fa)
// with F(x) = W phi(x) + bias, a smooth map R -> R^C.  F_d and F_a are tabulated once per set of weights on a uniform grid
// (step h = 1 / inv_h; row j <-> x = (j - 1) h; fp64 on the caller's side) and evaluated by 4-point Lagrange interpolation:
// error <= 0.024 h^4 max|d4F/dx4| ~ 2e-8 at h = 1/32 -- far below the fp32 rounding of the reference's own GEMM.  The
// 308 GFLOP per cloud become 16 coalesced 1 KB row reads (L2-resident tables) per (a, b) pair: the kernel is bound by the
// 4 N^2 C bytes it writes.  An index outside the table (d > table range) is evaluated directly from W, exactly.
constexpr int GT_T = 256;

__device__ __forceinline__ float4 ft_direct(const float* __restrict__ w, const float* __restrict__ b,
                                             const float* __restrict__ div_term, int C, int ch, float x) {
  float acc[4] = {b[ch], b[ch + 1], b[ch + 2], b[ch + 3]};
  for (int i = 0; i < C / 2; ++i) {
    const float om = x * div_term[i];
    const float sn = sinf(om), cs = cosf(om);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[(int64_t)(ch + j) * C + 2 * i + 1], cs, fmaf(w[(int64_t)(ch + j) * C + 2 * i], sn, acc[j]));
  }
  return make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__device__ __forceinline__ float4 ft_eval(const float4* __restrict__ tab, int rows, int c4, int cg, float inv_h, float x,
                                          const float* __restrict__ w, const float* __restrict__ b,
                                          const float* __restrict__ div_term) {
  const float m = floorf(x * inv_h);
  const float t = fmaf(x, inv_h, -m);  // one rounding: the position inside the cell keeps full precision
  if (!(m >= 0.0f) || !(m + 3.0f <= (float)(rows - 1))) return ft_direct(w, b, div_term, 4 * c4, 4 * cg, x);  // wave-uniform
  const float4* r = tab + (int64_t)(int)m * c4 + cg;
  const float4 v0 = r[0], v1 = r[c4], v2 = r[2 * c4], v3 = r[3 * c4];
  const float tm1 = t - 1.0f, tm2 = t - 2.0f, tp1 = t + 1.0f;
  const float w0 = -t * tm1 * tm2 * (1.0f / 6.0f), w1 = tp1 * tm1 * tm2 * 0.5f, w2 = -tp1 * t * tm2 * 0.5f,
              w3 = tp1 * t * tm1 * (1.0f / 6.0f);
  return make_float4(fmaf(w3, v3.x, fmaf(w2, v2.x, fmaf(w1, v1.x, w0 * v0.x))), fmaf(w3, v3.y, fmaf(w2, v2.y, fmaf(w1, v1.y, w0 * v0.y))),
                     fmaf(w3, v3.z, fmaf(w2, v2.z, fmaf(w1, v1.z, w0 * v0.z))), fmaf(w3, v3.w, fmaf(w2, v2.w, fmaf(w1, v1.w, w0 * v0.w))));
}

__global__ __launch_bounds__(GT_T) void geo_embedding_table_kernel(
    const float* __restrict__ pts, int n, const int32_t* __restrict__ knn, int k, const float4* __restrict__ tab_d, int rows_d,
    const float4* __restrict__ tab_a, int rows_a, float inv_h, int C, const float* __restrict__ w_d, const float* __restrict__ b_d,
    const float* __restrict__ w_a, const float* __restrict__ b_a, const float* __restrict__ div_term, float sigma_d,
    float factor_a, int mean, float* __restrict__ out) {
  __shared__ float xs[GE_KMAX + 1][GE_ROWS];  // [0..k-1] angular indices, [k] distance index
  const int64_t total = (int64_t)n * n;
  const int64_t r0 = (int64_t)blockIdx.x * GE_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  ge_pair_indices(pts, n, knn, k, sigma_d, factor_a, r0, total, tid, xs);
  __syncthreads();
  const int c4 = C / 4;
  float4* o4 = reinterpret_cast<float4*>(out);
  for (int r = w; r < GE_ROWS && r0 + r < total; r += GT_T / 64) {  // a wave per (a, b) pair; lanes over the channels
    const float xd = xs[k][r];
    for (int cg = lane; cg < c4; cg += 64) {
      float4 acc = ft_eval(tab_d, rows_d, c4, cg, inv_h, xd, w_d, b_d, div_term);
      if (k > 0) {
        float4 red = ft_eval(tab_a, rows_a, c4, cg, inv_h, xs[0][r], w_a, b_a, div_term);
        for (int i = 1; i < k; ++i) {
          const float4 v = ft_eval(tab_a, rows_a, c4, cg, inv_h, xs[i][r], w_a, b_a, div_term);
          if (mean) {
            red.x += v.x, red.y += v.y, red.z += v.z, red.w += v.w;
          } else {
            red.x = fmaxf(red.x, v.x), red.y = fmaxf(red.y, v.y), red.z = fmaxf(red.z, v.z), red.w = fmaxf(red.w, v.w);
          }
        }
        if (mean) {
          const float inv = 1.0f / (float)k;
          red.x *= inv, red.y *= inv, red.z *= inv, red.w *= inv;
        }
        acc.x += red.x, acc.y += red.y, acc.z += red.z, acc.w += red.w;
      }
      o4[(r0 + r) * c4 + cg] = acc;
    }
  }
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" int gr_geo_embedding_table(const float* points, int64_t n, const float* tab_d, int64_t rows_d, const float* tab_a,
                                      int64_t rows_a, float inv_h, const float* w_d, const float* b_d, const float* w_a,
                                      const float* b_a, const float* div_term, int64_t c, float sigma_d, float factor_a,
                                      int64_t angle_k, int reduction_mean, float* out, void* ws, size_t ws_bytes,
                                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && n < 46341, "geo_embedding: n*n must fit int32 pair ids per row (n=%lld)", (long long)n);
  GR_REQUIRE(c > 0 && c % 4 == 0, "geo_embedding_table: hidden_dim must be a positive multiple of 4 (got %lld)", (long long)c);
  GR_REQUIRE(angle_k >= 0 && angle_k <= GE_KMAX, "geo_embedding: angle_k must be in [0, %d]", GE_KMAX);
  GR_REQUIRE(angle_k < n || n == 0, "geo_embedding: angle_k (%lld) needs more than %lld points", (long long)angle_k, (long long)n);
  if (n == 0) return GR_OK;
  GR_REQUIRE(points && tab_d && w_d && b_d && div_term && out && (angle_k == 0 || (tab_a && w_a && b_a)), "null argument");
  GR_REQUIRE(rows_d >= 4 && (angle_k == 0 || rows_a >= 4) && inv_h > 0.0f, "geo_embedding_table: bad tables");
  GR_REQUIRE(reinterpret_cast<uintptr_t>(tab_d) % 16 == 0 && reinterpret_cast<uintptr_t>(tab_a) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(out) % 16 == 0, "geo_embedding_table: unaligned tensors");
  if (!ws || ws_bytes < gr_geo_embedding_workspace_bytes(n, angle_k)) {
    set_error("geo_embedding workspace too small");
    return GR_ERR_WORKSPACE;
  }
  Carver cv(ws);
  int32_t* knn = cv.take<int32_t>((size_t)n * std::max<int64_t>(angle_k, 1));
  if (angle_k > 0)
    hipLaunchKernelGGL(geo_knn_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, points, (int)n, (int)angle_k, knn);
  KernelTimer timer("geo_embedding", stream);
  hipLaunchKernelGGL(geo_embedding_table_kernel, dim3((unsigned)((n * n + GE_ROWS - 1) / GE_ROWS)), dim3(GT_T), 0, stream, points,
                     (int)n, knn, (int)angle_k, reinterpret_cast<const float4*>(tab_d), (int)rows_d,
                     reinterpret_cast<const float4*>(tab_a), (int)rows_a, inv_h, (int)c, w_d, b_d, w_a, b_a, div_term, sigma_d,
                     factor_a, reduction_mean & 1, out);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

// ---------------------------------------------------------------- RPE attention: positional score term
// rpe_transformer.py:55-57 projects the whole (N,M,C) embedding through proj_p in EVERY attention layer
// (2*N*M*C^2 flop = 77 GFLOP at N=M=767, C=256, plus a 602 MB temporary) and then contracts it with q:
//     s_p[h,n,m] = sum_c q[h,n,c] * (W_p emb[n,m] + b_p)[h*ch + c]
// The sum is linear in emb, so it is re-associated as   s_p[h,n,m] = emb[n,m,:] . u[n,h,:] + q[h,n,:].b_p[h]
// with u[n,h,:] = W_p[h-block]^T q[h,n,:] (a tiny GEMM done by the caller): one pass over the embedding, memory-bound.
// Workgroup = one query row n, 4 waves; 16 lanes share one (n,m) row (float4 loads, 256 B per 16 lanes).
namespace gr {
namespace {

template <int H, int CV>  // CV = C / 64 (float4 chunks per lane)
__global__ __launch_bounds__(256) void rpe_scores_kernel(const float* __restrict__ emb, const float* __restrict__ u,
                                                         const float* __restrict__ add, int n_rows, int m_cols,
                                                         float* __restrict__ out) {
  constexpr int C = CV * 64;
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane & 15, grp = lane >> 4;
  float4 ur[H][CV];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int i = 0; i < CV; ++i)
      ur[h][i] = *reinterpret_cast<const float4*>(u + ((int64_t)n * H + h) * C + i * 64 + sub * 4);
  float bias[H];
#pragma unroll
  for (int h = 0; h < H; ++h) bias[h] = add ? add[n * H + h] : 0.f;
  const int m_lo = blockIdx.y * 256;
  const int m_hi = min(m_lo + 256, m_cols);
  for (int m = m_lo + w * 4 + grp; m < m_hi; m += 16) {  // the 16 lanes of a row enter and leave together
    const bool live = true;
    const float* row = emb + ((int64_t)n * m_cols + m) * C + sub * 4;
    float4 e[CV];
#pragma unroll
    for (int i = 0; i < CV; ++i) e[i] = *reinterpret_cast<const float4*>(row + i * 64);
    float acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < CV; ++i) {
        a = fmaf(e[i].x, ur[h][i].x, a);
        a = fmaf(e[i].y, ur[h][i].y, a);
        a = fmaf(e[i].z, ur[h][i].z, a);
        a = fmaf(e[i].w, ur[h][i].w, a);
      }
      acc[h] = a;
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) acc[h] += __shfl_xor(acc[h], d, 64);
    }
    if (live && sub < H) {
      float v = acc[0];
#pragma unroll
      for (int h = 1; h < H; ++h) v = sub == h ? acc[h] : v;
      float bv = bias[0];
#pragma unroll
      for (int h = 1; h < H; ++h) bv = sub == h ? bias[h] : bv;
      out[((int64_t)sub * n_rows + n) * m_cols + m] = v + bv;
    }
  }
}

}  // namespace
}  // namespace gr

namespace gr {
namespace {

// ---------------------------------------------------------------- RPE attention, fused (rpe_transformer.py:51-72)
// One workgroup per query row n does the whole attention row for all heads:
//   1. raw scores  s[h][m] = (q[h,n,:] . k[h,m,:] + emb[n,m,:] . u[n,h,:] + add[n,h]) / sqrt(ch)   (16 lanes share one
//      (n, m) pair: float4 loads of the embedding row -- the only N*M*C stream, read exactly once per layer -- and of the
//      key row, xor-shuffle reduction), then attention_factors, key_weights, key_masks exactly in the reference's order;
//   2. softmax over m per head in LDS (wave reductions), written out as attention_scores (H, N, M);
//   3. hidden[n, h*ch + c] = sum_m p[h][m] * v[m, h*ch + c]  (thread = output channel, the value matrix streams from L2).
// Nothing of size N*M*C or H*N*M is re-read from HBM between the steps; the reference materialises the (N, M, C) projected
// embedding, two (H, N, M) score tensors and the softmax in separate ATen kernels.
template <int H, int CV>
__global__ __launch_bounds__(256) void rpe_attention_kernel(const float* __restrict__ emb, const float* __restrict__ u,
                                                            const float* __restrict__ add, const float* __restrict__ q,
                                                            const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ factors,
                                                            const float* __restrict__ key_weights,
                                                            const uint8_t* __restrict__ key_masks, int n_rows, int m_cols,
                                                            float inv_sqrt_ch, float* __restrict__ out_scores,
                                                            float* __restrict__ out_hidden) {
  constexpr int C = CV * 64, CH = C / H;
  extern __shared__ float s_sc[];  // [H][m_cols]
  __shared__ float s_red[2][8];
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, sub = lane & 15, grp = lane >> 4;
  float4 ur[H][CV], qr[CV];
#pragma unroll
  for (int i = 0; i < CV; ++i) {
    qr[i] = *reinterpret_cast<const float4*>(q + (int64_t)n * C + i * 64 + sub * 4);
#pragma unroll
    for (int h = 0; h < H; ++h) ur[h][i] = *reinterpret_cast<const float4*>(u + ((int64_t)n * H + h) * C + i * 64 + sub * 4);
  }
  // ---- 1. raw scores
  for (int m = w * 4 + grp; m < m_cols; m += 16) {
    const float* erow = emb + ((int64_t)n * m_cols + m) * C + sub * 4;
    const float* krow = k + (int64_t)m * C + sub * 4;
    float4 e[CV], kk[CV];
#pragma unroll
    for (int i = 0; i < CV; ++i) {
      e[i] = *reinterpret_cast<const float4*>(erow + i * 64);
      kk[i] = *reinterpret_cast<const float4*>(krow + i * 64);
    }
    float acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < CV; ++i) {
        a = fmaf(e[i].x, ur[h][i].x, a);
        a = fmaf(e[i].y, ur[h][i].y, a);
        a = fmaf(e[i].z, ur[h][i].z, a);
        a = fmaf(e[i].w, ur[h][i].w, a);
      }
      acc[h] = a;
    }
#pragma unroll
    for (int i = 0; i < CV; ++i) {  // q . k: the four channels of this float4 belong to head (i*64 + sub*4) / CH
      const float part = fmaf(qr[i].x, kk[i].x, fmaf(qr[i].y, kk[i].y, fmaf(qr[i].z, kk[i].z, qr[i].w * kk[i].w)));
      const int hd = (i * 64 + sub * 4) / CH;
#pragma unroll
      for (int h = 0; h < H; ++h) acc[h] += hd == h ? part : 0.f;
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) acc[h] += __shfl_xor(acc[h], d, 64);
    }
    if (sub < H) {
      float val = acc[0];
#pragma unroll
      for (int h = 1; h < H; ++h) val = sub == h ? acc[h] : val;
      float sc = (val + add[n * H + sub]) * inv_sqrt_ch;
      if (factors) sc = factors[(int64_t)n * m_cols + m] * sc;
      if (key_weights) sc = sc * key_weights[m];
      if (key_masks && key_masks[m]) sc = -INFINITY;
      s_sc[sub * m_cols + m] = sc;
    }
  }
  __syncthreads();
  // ---- 2. softmax over m, one wave per head (two heads per wave when H = 8)
  for (int h = w; h < H; h += 4) {
    float* row = s_sc + h * m_cols;
    float mx = -INFINITY;
    for (int m = lane; m < m_cols; m += 64) mx = fmaxf(mx, row[m]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float sum = 0.f;
    for (int m = lane; m < m_cols; m += 64) {
      const float ev = expf(row[m] - mx);
      row[m] = ev;
      sum += ev;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    const float inv = 1.0f / sum;
    float* dst = out_scores + ((int64_t)h * n_rows + n) * m_cols;
    for (int m = lane; m < m_cols; m += 64) {
      const float p = row[m] * inv;
      row[m] = p;
      dst[m] = p;
    }
  }
  __syncthreads();
  // ---- 3. hidden = P V: wave w takes a quarter of the keys, lane l the four channels 4l .. 4l+3 (float4 loads of the
  //         value rows, eight keys in flight), then the four partial rows are added through LDS
  __shared__ float4 s_part[4][64];
  {
    const int per = (m_cols + 3) / 4;
    const int m0 = w * per, m1 = min(m_cols, m0 + per);
    const bool on = lane * 4 < C;
    const float* prow = s_sc + ((lane * 4) / CH) * m_cols;
    const float* vcol = v + lane * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
      int m = m0;
      for (; m + 8 <= m1; m += 8) {
        float4 vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = *reinterpret_cast<const float4*>(vcol + (int64_t)(m + j) * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = prow[m + j];
          acc.x = fmaf(p, vv[j].x, acc.x);
          acc.y = fmaf(p, vv[j].y, acc.y);
          acc.z = fmaf(p, vv[j].z, acc.z);
          acc.w = fmaf(p, vv[j].w, acc.w);
        }
      }
      for (; m < m1; ++m) {
        const float4 vv = *reinterpret_cast<const float4*>(vcol + (int64_t)m * C);
        const float p = prow[m];
        acc.x = fmaf(p, vv.x, acc.x);
        acc.y = fmaf(p, vv.y, acc.y);
        acc.z = fmaf(p, vv.z, acc.z);
        acc.w = fmaf(p, vv.w, acc.w);
      }
    }
    s_part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && on) {
      const float4 a = s_part[0][lane], b2 = s_part[1][lane], c2 = s_part[2][lane], d2 = s_part[3][lane];
      float4 r;
      r.x = (a.x + b2.x) + (c2.x + d2.x);
      r.y = (a.y + b2.y) + (c2.y + d2.y);
      r.z = (a.z + b2.z) + (c2.z + d2.z);
      r.w = (a.w + b2.w) + (c2.w + d2.w);
      *reinterpret_cast<float4*>(out_hidden + (int64_t)n * C + lane * 4) = r;
    }
  }
  (void)s_red;
}

}  // namespace
}  // namespace gr

extern "C" int gr_rpe_attention(const float* embed, const float* u, const float* add, const float* q, const float* k,
                                const float* v, const float* attention_factors, const float* key_weights,
                                const uint8_t* key_masks, int64_t n, int64_t m, int64_t c, int64_t heads, float* out_scores,
                                float* out_hidden, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && m >= 0 && n < (1 << 24) && m < (1 << 24), "rpe_attention: bad sizes");
  GR_REQUIRE((c == 64 || c == 128 || c == 256) && (heads == 1 || heads == 2 || heads == 4 || heads == 8),
             "rpe_attention: d_model must be 64/128/256 and num_heads 1/2/4/8 (got %lld, %lld)", (long long)c, (long long)heads);
  if (n == 0) return GR_OK;
  GR_REQUIRE(m > 0, "rpe_attention: no keys (softmax over an empty row)");
  GR_REQUIRE(embed && u && add && q && k && v && out_scores && out_hidden, "null argument");
  const size_t lds = (size_t)heads * m * sizeof(float);
  GR_REQUIRE(lds <= 150 * 1024, "rpe_attention: %lld keys x %lld heads do not fit in LDS", (long long)m, (long long)heads);
  const float inv_sqrt_ch = 1.0f / sqrtf((float)(c / heads));
  KernelTimer timer("rpe_attention", stream);
#define GR_RPA(H, CV)                                                                                            \
  do {                                                                                                           \
    auto kern = gr::rpe_attention_kernel<H, CV>;                                                                 \
    if (lds > 64 * 1024)                                                                                         \
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 160 * 1024));                                                                   \
    hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(256), lds, stream, embed, u, add, q, k, v, attention_factors, \
                       key_weights, key_masks, (int)n, (int)m, inv_sqrt_ch, out_scores, out_hidden);             \
  } while (0)
#define GR_RPA_H(CV)               \
  switch (heads) {                 \
    case 1: GR_RPA(1, CV); break;  \
    case 2: GR_RPA(2, CV); break;  \
    case 4: GR_RPA(4, CV); break;  \
    default: GR_RPA(8, CV); break; \
  }
  if (c == 64) { GR_RPA_H(1); } else if (c == 128) { GR_RPA_H(2); } else { GR_RPA_H(4); }
#undef GR_RPA_H
#undef GR_RPA
  GR_LAUNCH_CHECK();
  return GR_OK;
}

extern "C" int gr_rpe_scores(const float* embed, const float* u, const float* add, int64_t n, int64_t m, int64_t c,
                             int64_t heads, float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && m >= 0 && n < (1 << 24) && m < (1 << 24), "rpe_scores: bad sizes");
  GR_REQUIRE((c == 64 || c == 128 || c == 256) && (heads == 1 || heads == 2 || heads == 4 || heads == 8),
             "rpe_scores: d_model must be 64/128/256 and num_heads 1/2/4/8 (got %lld, %lld)", (long long)c, (long long)heads);
  if (n == 0 || m == 0) return GR_OK;
  GR_REQUIRE(embed && u && out, "null argument");
  const dim3 grid((unsigned)n, (unsigned)((m + 255) / 256));
  KernelTimer timer("rpe_scores", stream);
#define GR_RPE(H, CV)                                                                                          \
  hipLaunchKernelGGL((rpe_scores_kernel<H, CV>), grid, dim3(256), 0, stream, embed, u, add, (int)n, (int)m, out)
#define GR_RPE_H(CV)          \
  switch (heads) {            \
    case 1: GR_RPE(1, CV); break; \
    case 2: GR_RPE(2, CV); break; \
    case 4: GR_RPE(4, CV); break; \
    default: GR_RPE(8, CV); break; \
  }
  if (c == 64) { GR_RPE_H(1) } else if (c == 128) { GR_RPE_H(2) } else { GR_RPE_H(4) }
#undef GR_RPE_H
#undef GR_RPE
  GR_LAUNCH_CHECK();
  return GR_OK;
}
