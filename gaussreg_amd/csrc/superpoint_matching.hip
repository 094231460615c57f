// pairwise_distance (the one dense contraction on the path, fp32 MFMA) and SuperPointMatching.
//
// Replaces the ATen call chains of
//   geotransformer/modules/ops/pairwise_distance.py:4-31                       (a8)
//   geotransformer/modules/geotransformer/superpoint_matching.py:32-48          (a9)
// pairwise: d = clamp((x2 - 2*xy) + y2, 0)  or  clamp(2 - 2*xy, 0) with xy on v_mfma_f32_32x32x2_f32
// (exact fp32, == an fmaf chain over k -- cdna_hip_programming.md section 3).
// SuperPointMatching: mask compaction -> S = exp(-d) -> row / column sums (fixed summation order,
// no float atomics: results are reproducible run to run) -> score = (S/rowsum)*(S/colsum) ->
// exact global top-k by a 3-pass radix select on the score bits (ties: lowest flat index first)
// -> sorted (score desc) -> indices mapped back through the compaction tables.
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace gr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PD_T = 64;   // output tile 64 x 64 per 256-thread block, 32 x 32 per wave
constexpr int PD_K = 32;   // k-slab staged in LDS
constexpr int PD_LD = PD_K + 1;

enum { EPI_DIST = 0, EPI_EXPNEG = 1 };

// Stack mode (gr_superpoint_matching_batch): the kernels of one pair, launched once for all pairs of a batch.  Pair z
// (blockIdx.z, or blockIdx.x for the single-workgroup kernels) finds its superpoints in the stacked inputs through
// `stack[z]`; its workspace is the one the pointer arguments name, `zstride` bytes further per pair.
struct SpmStack {
  int32_t r0, nr, s0, ns;  // first ref / src superpoint in the stack, their counts
};
// plain batches (pairwise_distance on (B, N, C) x (B, M, C)): element strides from one matrix of the batch to the next
struct PdBatch {
  int64_t x, y, out, x2, y2;
};
template <class T>
__device__ __forceinline__ T* z_shift(T* p, size_t bytes) {  // (pointer arithmetic, not integers: the address space stays known)
  return p ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + bytes) : p;
}

// SuperPointMatching's normalisation needs the row and column sums of the exp(-d) matrix: the distance kernels leave
// PARTIAL sums of their output tile -- rs_part[tile column][row] = the row's sum over the tile's columns, cs_part[tile row]
// [column] = the column's sum over the tile's rows -- and sums_finish_kernel adds a row's / column's partials in tile order:
// fixed summation order, no float atomics, and nobody reads the matrix again for it (the two-kernel form read it twice).
// v[a][b][r]: the wave's values in MFMA C/D layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) inside the
// 32 x 32 block (a, b)), zero outside the matrix.  WR x WC waves per workgroup; wr / wc = this wave's position; s_red: LDS
// scratch of (WR + WC) * T floats (T = 32 * max(A * WR, B * WC) = the tile edge), free when this is called.
template <int A, int B, int WR, int WC>
__device__ __forceinline__ void tile_partial_sums(const float (&v)[A][B][16], int wr, int wc, int lane, int tid, float* s_red,
                                                  int i0, int j0, int n, int m, float* __restrict__ rs_row,
                                                  float* __restrict__ cs_row) {
  constexpr int T = 32 * A * WR;
  static_assert(A * WR == B * WC, "square workgroup tiles");
  float* s_rows = s_red;            // [WC][T]
  float* s_cols = s_red + WC * T;   // [WR][T]
  // rows: over b, then a butterfly over the 32 lanes of the half-wave (every lane ends with the sum)
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = v[a][0][r];
#pragma unroll
      for (int b = 1; b < B; ++b) t += v[a][b][r];
#pragma unroll
      for (int d = 16; d > 0; d >>= 1) t += __shfl_xor(t, d, WAVE);
      if ((lane & 31) == 0) s_rows[wc * T + wr * 32 * A + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = t;
    }
  // columns: the lane's 16 A values, then the other half-wave's
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += v[a][b][r];
    t += __shfl_xor(t, 32, WAVE);
    if (lane < 32) s_cols[wr * T + wc * 32 * B + 32 * b + lane] = t;
  }
  __syncthreads();
  if (tid < T) {
    float t = s_rows[tid];
#pragma unroll
    for (int w = 1; w < WC; ++w) t += s_rows[w * T + tid];
    if (i0 + tid < n) rs_row[i0 + tid] = t;
    float u = s_cols[tid];
#pragma unroll
    for (int w = 1; w < WR; ++w) u += s_cols[w * T + tid];
    if (j0 + tid < m) cs_row[j0 + tid] = u;
  }
}

// out[i][j] = epilogue(dist(x[xi[i]], y[yi[j]]));  xi / yi optional gather tables (nullptr = identity).
// n_dev / m_dev (optional) hold the row / column counts on the device (after a compaction).
template <int EPI>
__global__ __launch_bounds__(256) void pairwise_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const int32_t* __restrict__ xi,
    const int32_t* __restrict__ yi, int n, int m, const int32_t* __restrict__ nm_dev, int C,
    int normalized, const float* __restrict__ x2, const float* __restrict__ y2,
    float* __restrict__ out, int ld_out, const SpmStack* __restrict__ stack, size_t zstride, PdBatch bs,
    float* __restrict__ rs_part, float* __restrict__ cs_part, int ld_rs, int ld_cs) {
  __shared__ float sx[PD_T][PD_LD];
  __shared__ float sy[PD_T][PD_LD];
  if (stack) {  // x = y = the stacked features; the gather tables, counts and the output belong to pair blockIdx.z
    const SpmStack P = stack[blockIdx.z];
    y = x + (int64_t)P.s0 * C;
    x = x + (int64_t)P.r0 * C;
    n = P.nr;
    m = P.ns;
    ld_out = P.ns;
    const size_t zo = (size_t)blockIdx.z * zstride;
    xi = z_shift(xi, zo), yi = z_shift(yi, zo), nm_dev = z_shift(nm_dev, zo), out = z_shift(out, zo);
    rs_part = z_shift(rs_part, zo), cs_part = z_shift(cs_part, zo);
  }
  if (!stack && gridDim.z > 1) {  // plain batch: matrix blockIdx.z
    const int64_t z = blockIdx.z;
    x += z * bs.x, y += z * bs.y, out += z * bs.out;
    if (x2) x2 += z * bs.x2, y2 += z * bs.y2;
  }
  if (nm_dev) {
    n = nm_dev[0];
    m = nm_dev[1];
  }
  const int i0 = blockIdx.y * PD_T, j0 = blockIdx.x * PD_T;
  if (i0 >= n || j0 >= m) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32;  // wave's 32x32 sub-tile
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // staging: element e of a 64 x 32 slab = (row e / 32, k e % 32); a thread owns eight of them, the same (row, k) pair of
  // every slab, so its row addresses are fixed for the whole k loop.  The loads of slab s + 1 are issued before the MFMAs of
  // slab s (registers) and land while they run: one global round trip per slab used to sit between two barriers with nothing
  // to cover it (144 workgroups of a 767 x 767 call: 34 us for 0.3 GFLOP)
  constexpr int PER = PD_T * PD_K / 256;  // 8
  const float* xrow[PER];
  const float* yrow[PER];
  bool xok[PER], yok[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = tid + u * 256, r = e / PD_K;
    const int gi = i0 + r, gj = j0 + r;
    xok[u] = gi < n;
    yok[u] = gj < m;
    xrow[u] = x + (int64_t)(xok[u] ? (xi ? xi[gi] : gi) : 0) * C + (e % PD_K);
    yrow[u] = y + (int64_t)(yok[u] ? (yi ? yi[gj] : gj) : 0) * C + (e % PD_K);
  }
  float rx[PER], ry[PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const bool kin = k0 + ((tid + u * 256) % PD_K) < C;
      rx[u] = (xok[u] && kin) ? xrow[u][k0] : 0.f;   // out-of-range -> 0
      ry[u] = (yok[u] && kin) ? yrow[u][k0] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < C; k0 += PD_K) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 256;
      sx[e / PD_K][e % PD_K] = rx[u];
      sy[e / PD_K][e % PD_K] = ry[u];
    }
    __syncthreads();
    if (k0 + PD_K < C) fetch(k0 + PD_K);
#pragma unroll
    for (int k = 0; k < PD_K; k += 2) {
      // A[i = lane & 31][k = lane >> 5],  B[k = lane >> 5][j = lane & 31]
      const float a = sx[wi + (lane & 31)][k + (lane >> 5)];
      const float b = sy[wj + (lane & 31)][k + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float vals[1][1][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gi = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int gj = j0 + wj + (lane & 31);
    vals[0][0][r] = 0.f;
    if (gi < n && gj < m) {
      const float xy = acc[r];
      float d;
      if (normalized) d = 2.0f - 2.0f * xy;                    // pairwise_distance.py:26
      else d = (x2[gi] - 2.0f * xy) + y2[gj];                   // pairwise_distance.py:30
      d = fmaxf(d, 0.0f);                                       // :31 clamp(min=0)
      if (EPI == EPI_EXPNEG) d = expf(-d);                      // superpoint_matching.py:37
      out[(int64_t)gi * ld_out + gj] = d;
      vals[0][0][r] = d;
    }
  }
  if (EPI == EPI_EXPNEG && rs_part != nullptr)  // (the k loop ended with a barrier: the staging slabs are free)
    tile_partial_sums<1, 1, 2, 2>(vals, w >> 1, w & 1, lane, tid, &sx[0][0], i0, j0, n, m, rs_part + (int64_t)blockIdx.x * ld_rs,
                                  cs_part + (int64_t)blockIdx.y * ld_cs);
}

// The same contraction for LARGE outputs (and for the batched SuperPointMatching launch): 128 x 128 per workgroup, 64 x 64 per
// wave = four 32 x 32 accumulators, so one pair of LDS reads feeds two MFMAs instead of one (the 64 x 64 kernel above issues
// two ds_read_b32 per v_mfma_f32_32x32x2_f32 and waits for a global round trip per k-slab with nothing to cover it: 56 TF at
// 8192^2); k-slabs of 16 through TWO LDS buffers: the global loads of slab s + 1 are issued before the MFMAs of slab s and
// land in registers while they run, one barrier per slab.  Same arithmetic per output element (an fmaf chain over k in
// ascending order), so the two kernels return the same bits.
constexpr int PB_T = 128;  // output tile per workgroup
constexpr int PB_K = 16;   // k-slab
constexpr int PB_LD = PB_K + 1;

// ALIGNED: C is a multiple of the k-slab and both matrices are 16-byte aligned -- every fetch is one unconditional float4
// load on a clamped row (behind a branch the compiler waits for each load before it issues the next: four memory round
// trips per slab instead of one).
template <int EPI, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void pairwise_big_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const int32_t* __restrict__ xi,
    const int32_t* __restrict__ yi, int n, int m, const int32_t* __restrict__ nm_dev, int C,
    int normalized, const float* __restrict__ x2, const float* __restrict__ y2,
    float* __restrict__ out, int ld_out, const SpmStack* __restrict__ stack, size_t zstride, PdBatch bs,
    float* __restrict__ rs_part, float* __restrict__ cs_part, int ld_rs, int ld_cs) {
  __shared__ float sx[2][PB_T][PB_LD];
  __shared__ float sy[2][PB_T][PB_LD];
  if (stack) {
    const SpmStack P = stack[blockIdx.z];
    y = x + (int64_t)P.s0 * C;
    x = x + (int64_t)P.r0 * C;
    n = P.nr;
    m = P.ns;
    ld_out = P.ns;
    const size_t zo = (size_t)blockIdx.z * zstride;
    xi = z_shift(xi, zo), yi = z_shift(yi, zo), nm_dev = z_shift(nm_dev, zo), out = z_shift(out, zo);
    rs_part = z_shift(rs_part, zo), cs_part = z_shift(cs_part, zo);
  }
  if (!stack && gridDim.z > 1) {  // plain batch: matrix blockIdx.z
    const int64_t z = blockIdx.z;
    x += z * bs.x, y += z * bs.y, out += z * bs.out;
    if (x2) x2 += z * bs.x2, y2 += z * bs.y2;
  }
  if (nm_dev) {
    n = nm_dev[0];
    m = nm_dev[1];
  }
  const int i0 = blockIdx.y * PB_T, j0 = blockIdx.x * PB_T;
  if (i0 >= n || j0 >= m) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 1) * 64, wj = (w & 1) * 64;  // wave's 64 x 64 sub-tile
  // staging: thread -> (row r0 + 64 u, four consecutive k); the row's address is fixed for the whole k loop
  const int sr = tid >> 2, sk = (tid & 3) * 4;
  const float* xrow[2];
  const float* yrow[2];
  bool xok[2], yok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int gi = i0 + sr + 64 * u, gj = j0 + sr + 64 * u;
    xok[u] = gi < n;
    yok[u] = gj < m;
    xrow[u] = x + (int64_t)(xok[u] ? (xi ? xi[gi] : gi) : 0) * C;
    yrow[u] = y + (int64_t)(yok[u] ? (yi ? yi[gj] : gj) : 0) * C;
  }
  const bool vec = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0);
  auto fetch = [&](const float* row, bool ok, int k0) -> float4 {
    const int gk = k0 + sk;
    if (ALIGNED) {
      const float4 t = *reinterpret_cast<const float4*>(row + gk);  // rows out of range read row 0 and are zeroed here
      return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      if (vec && gk + 4 <= C) {
        v = *reinterpret_cast<const float4*>(row + gk);
      } else {
        if (gk < C) v.x = row[gk];
        if (gk + 1 < C) v.y = row[gk + 1];
        if (gk + 2 < C) v.z = row[gk + 2];
        if (gk + 3 < C) v.w = row[gk + 3];
      }
    }
    return v;
  };
  auto stash = [&](float (*dst)[PB_LD], int u, const float4 v) {
    float* d = &dst[sr + 64 * u][sk];
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float4 rx[2], ry[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    rx[u] = fetch(xrow[u], xok[u], 0);
    ry[u] = fetch(yrow[u], yok[u], 0);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    stash(sx[0], u, rx[u]);
    stash(sy[0], u, ry[u]);
  }
  __syncthreads();
  const int slabs = (C + PB_K - 1) / PB_K;
  for (int sidx = 0; sidx < slabs; ++sidx) {
    const int cur = sidx & 1;
    const bool more = sidx + 1 < slabs;
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        rx[u] = fetch(xrow[u], xok[u], (sidx + 1) * PB_K);
        ry[u] = fetch(yrow[u], yok[u], (sidx + 1) * PB_K);
      }
    }
#pragma unroll
    for (int k = 0; k < PB_K; k += 2) {
      // A[i = lane & 31][k = lane >> 5],  B[k = lane >> 5][j = lane & 31]
      const int kk = k + (lane >> 5), rr = lane & 31;
      const float a0 = sx[cur][wi + rr][kk], a1 = sx[cur][wi + 32 + rr][kk];
      const float b0 = sy[cur][wj + rr][kk], b1 = sy[cur][wj + 32 + rr][kk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        stash(sx[cur ^ 1], u, rx[u]);
        stash(sy[cur ^ 1], u, ry[u]);
      }
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // the squared norms of the lane's rows and columns, requested up front on clamped indices (one memory round trip)
  float xx[2][16], yy[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gi = i0 + wi + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      xx[a][r] = normalized ? 0.f : x2[min(gi, n - 1)];
    }
#pragma unroll
  for (int b = 0; b < 2; ++b) yy[b] = normalized ? 0.f : y2[min(j0 + wj + 32 * b + (lane & 31), m - 1)];
  float vals[2][2][16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int gj = j0 + wj + 32 * b + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gi = i0 + wi + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float xy = acc[a][b][r];
        float d;
        if (normalized) d = 2.0f - 2.0f * xy;      // pairwise_distance.py:26
        else d = (xx[a][r] - 2.0f * xy) + yy[b];   // pairwise_distance.py:30
        d = fmaxf(d, 0.0f);                        // :31 clamp(min=0)
        if (EPI == EPI_EXPNEG) d = expf(-d);       // superpoint_matching.py:37
        const bool in = gi < n && gj < m;
        if (in) out[(int64_t)gi * ld_out + gj] = d;
        vals[a][b][r] = in ? d : 0.f;
      }
    }
  if (EPI == EPI_EXPNEG && rs_part != nullptr)  // (the k loop ended with a barrier: the staging slabs are free)
    tile_partial_sums<2, 2, 2, 2>(vals, w >> 1, w & 1, lane, tid, &sx[0][0][0], i0, j0, n, m, rs_part + (int64_t)blockIdx.x * ld_rs,
                                  cs_part + (int64_t)blockIdx.y * ld_cs);
}

// squared norms of the rows: one wave per row, lanes stride the channels (coalesced; a thread per row read its 1 KB row on
// its own: 64 separate lines per load instruction, ~45 us of the 8192 x 256 call), fixed-shape reduction
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, int n, int C,
                                                     float* __restrict__ out) {
  const int row = blockIdx.x * (256 / WAVE) + (int)(threadIdx.x / WAVE), lane = threadIdx.x & (WAVE - 1);
  if (row >= n) return;
  const float* r = x + (int64_t)row * C;
  float s = 0.f;
  for (int k = lane; k < C; k += WAVE) {
    const float v = r[k];
    s += v * v;
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) s += __shfl_xor(s, d, WAVE);
  if (lane == 0) out[row] = s;
}

// ---------------------------------------------------------------- SuperPointMatching pieces
struct SpmHdr {
  int32_t nr, ns;       // compacted sizes
  uint32_t prefix;      // selected high bits so far
  int32_t k_rem;        // how many still to take inside the selected bin
  int32_t k;            // min(num_correspondences, nr * ns)
  int32_t n_cand;       // candidates gathered (score >= threshold)
  uint32_t tau_max;     // fast path: the largest slab threshold -- no score below it can be among the k best of the pair
  int32_t overflow;     // fast path: a slab list, the candidate buffer or the sort buffer overflowed -> dense selection
};

// single block: order-preserving compaction of the true mask entries (torch.nonzero order)
__global__ __launch_bounds__(1024) void compact_masks_kernel(const uint8_t* __restrict__ rm, int nr_all,
                                                             const uint8_t* __restrict__ sm, int ns_all,
                                                             int num_corr, int32_t* __restrict__ ridx,
                                                             int32_t* __restrict__ sidx,
                                                             SpmHdr* __restrict__ hdr, uint32_t* __restrict__ hist,
                                                             const SpmStack* __restrict__ stack, size_t zstride) {
  __shared__ int wsum[1024 / WAVE];
  __shared__ int carry;
  if (stack) {  // rm = sm = the stacked masks (or null); workgroup = pair; also clears the pair's selection histograms
    const SpmStack P = stack[blockIdx.x];
    sm = sm ? sm + P.s0 : sm;
    rm = rm ? rm + P.r0 : rm;
    nr_all = P.nr;
    ns_all = P.ns;
    const size_t zo = (size_t)blockIdx.x * zstride;
    ridx = z_shift(ridx, zo), sidx = z_shift(sidx, zo), hdr = z_shift(hdr, zo), hist = z_shift(hist, zo);
    for (int i = threadIdx.x; i < 3 * 2048; i += 1024) hist[i] = 0u;
  }
  for (int which = 0; which < 2; ++which) {
    const uint8_t* mask = which ? sm : rm;
    const int n = which ? ns_all : nr_all;
    int32_t* dst = which ? sidx : ridx;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
      const int i = base + threadIdx.x;
      const int f = (i < n) && (mask == nullptr || mask[i] != 0);
      const unsigned long long bal = __ballot(f);
      const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
      if (lane == 0) wsum[w] = __popcll(bal);
      __syncthreads();
      int off = carry;
      for (int u = 0; u < w; ++u) off += wsum[u];
      if (f) dst[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
      __syncthreads();
      if (threadIdx.x == 0) {
        int t = 0;
        for (int u = 0; u < 1024 / WAVE; ++u) t += wsum[u];
        carry += t;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      if (which) hdr->ns = carry; else hdr->nr = carry;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long tot = (long long)hdr->nr * hdr->ns;
    hdr->k = (int)(tot < num_corr ? tot : num_corr);
    hdr->prefix = 0;
    hdr->k_rem = hdr->k;
    hdr->n_cand = 0;
    hdr->tau_max = 0u;
    hdr->overflow = 0;
  }
}

// row / column sums of the exp(-d) matrix from the partial sums the distance kernel left (tile_partial_sums): a row's partials
// in tile-column order, a column's in tile-row order -- fixed summation order
__global__ __launch_bounds__(256) void sums_finish_kernel(const SpmHdr* __restrict__ hdr, const float* __restrict__ rs_part,
                                                          const float* __restrict__ cs_part, int ld_rs, int ld_cs, int tile,
                                                          float* __restrict__ rs, float* __restrict__ cs,
                                                          const SpmStack* __restrict__ stack, size_t zstride) {
  if (stack) {
    const size_t zo = (size_t)blockIdx.z * zstride;
    hdr = z_shift(hdr, zo), rs_part = z_shift(rs_part, zo), cs_part = z_shift(cs_part, zo), rs = z_shift(rs, zo), cs = z_shift(cs, zo);
  }
  const int nr = hdr->nr, ns = hdr->ns;
  const int ntc = (ns + tile - 1) / tile, ntr = (nr + tile - 1) / tile;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nr) {
    float t = 0.f;
    for (int c = 0; c < ntc; ++c) t += rs_part[(int64_t)c * ld_rs + i];
    rs[i] = t;
  }
  if (i < ns) {
    float t = 0.f;
    for (int r = 0; r < ntr; ++r) t += cs_part[(int64_t)r * ld_cs + i];
    cs[i] = t;
  }
}

// scores, dense (nr x ns) row-major: superpoint_matching.py:38-41  (S / rowsum) * (S / colsum)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ S, int ld,
                                                        const float* __restrict__ rs,
                                                        const float* __restrict__ cs, int dual,
                                                        const SpmHdr* __restrict__ hdr,
                                                        float* __restrict__ score, const SpmStack* __restrict__ stack,
                                                        size_t zstride) {
  if (stack) {
    ld = stack[blockIdx.z].ns;
    const size_t zo = (size_t)blockIdx.z * zstride;
    S = z_shift(S, zo), rs = z_shift(rs, zo), cs = z_shift(cs, zo), hdr = z_shift(hdr, zo), score = z_shift(score, zo);
  }
  const int nr = hdr->nr, ns = hdr->ns;
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (r >= nr || c >= ns) return;
  const float s = S[(int64_t)r * ld + c];
  score[(int64_t)r * ns + c] = dual ? (s / rs[r]) * (s / cs[c]) : s;
}

// LDS histogram increment that stays fast when most lanes hit the same bin (the scores of one
// matrix share their exponent): lanes with equal bins are merged with ballots, one lane adds the count.
__device__ __forceinline__ void hist_add(uint32_t* sh, uint32_t bin, bool active) {
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lb = (uint32_t)__shfl((int)bin, leader, WAVE);
    const unsigned long long same = __ballot(active && bin == lb) & todo;
    if ((int)(threadIdx.x & (WAVE - 1)) == leader) atomicAdd(&sh[lb], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

constexpr int SEL_BITS[3] = {11, 11, 10};
constexpr int SEL_SHIFT[3] = {21, 10, 0};

// The k-th largest score is found by a 3-digit radix select (11 / 11 / 10 bits of the fp32 pattern).
// Every selection kernel first re-derives, from the previous digit's histogram, which bin holds the
// k-th largest (`pick`: a 2048-entry scan every block repeats for itself -- cheaper than a launch).
struct Pick {
  uint32_t prefix;
  int k_rem;
};

__device__ inline Pick pick_from_hist(const uint32_t* __restrict__ hist_prev, int prev_pass, uint32_t prefix,
                                      int k_rem, uint32_t* sh /* [2048] */) {
  // block-parallel (256 threads): thread t owns bins [8t, 8t+8); a suffix scan over the per-thread
  // totals finds the thread whose bins contain the k-th largest, which then walks its 8 bins.
  // (A single thread walking 2048 LDS bins costs ~100 us of dependent-load latency.)
  __shared__ Pick res;
  __shared__ uint32_t tsum[256];
  const int nb = 1 << SEL_BITS[prev_pass];
  const int t = threadIdx.x;
  uint32_t mine = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int bin = 8 * t + i;
    const uint32_t v = bin < nb ? hist_prev[bin] : 0;
    sh[bin] = v;
    mine += v;
  }
  tsum[t] = mine;
  __syncthreads();
  // inclusive suffix scan of tsum (Hillis-Steele, 8 steps)
  for (int d = 1; d < 256; d <<= 1) {
    const uint32_t add = (t + d < 256) ? tsum[t + d] : 0;
    __syncthreads();
    tsum[t] += add;
    __syncthreads();
  }
  if (t == 0) {  // default when k_rem == 0 or fewer than k_rem elements exist: lowest bin
    res.prefix = prefix;
    res.k_rem = k_rem;
  }
  __syncthreads();
  const uint32_t above = tsum[t] - mine;  // elements in bins of higher threads
  if (k_rem > 0 && above < (uint32_t)k_rem && (uint32_t)k_rem <= tsum[t]) {
    int k = k_rem - (int)above;
    int bsel = 8 * t;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
      const int c = (int)sh[8 * t + i];
      if (c >= k) {
        bsel = 8 * t + i;
        break;
      }
      k -= c;
    }
    res.prefix = prefix | ((uint32_t)bsel << SEL_SHIFT[prev_pass]);
    res.k_rem = k;
  } else if (k_rem == 0 && t == 0) {
    res.prefix = prefix | ((uint32_t)(nb - 1) << SEL_SHIFT[prev_pass]);
  }
  __syncthreads();
  return res;
}

// pass 0: plain histogram of the top digit.  pass 1/2: pick digit pass-1 first (from hist[pass-1]).
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ score,
                                                          SpmHdr* __restrict__ hdr, int pass,
                                                          uint32_t* __restrict__ hist /* [3][2048] */, size_t zstride) {
  __shared__ uint32_t sh[2048];
  {
    const size_t zo = (size_t)blockIdx.z * zstride;
    score = z_shift(score, zo), hdr = z_shift(hdr, zo), hist = z_shift(hist, zo);
  }
  uint32_t prefix = 0;
  int k_rem = hdr->k;
  for (int pp = 0; pp < pass; ++pp) {  // replay the picks of the earlier digits (deterministic)
    const Pick pk = pick_from_hist(hist + pp * 2048, pp, prefix, k_rem, sh);
    prefix = pk.prefix;
    k_rem = pk.k_rem;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 2048; i += 256) sh[i] = 0;
  __syncthreads();
  const int ns = hdr->ns;
  const int64_t total = (int64_t)hdr->nr * ns;
  const int shift = SEL_SHIFT[pass], bits = SEL_BITS[pass];
  const int hi_shift = shift + bits;
  const int64_t total_up = (total + 255) / 256 * 256;  // whole waves enter hist_add together
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_up; e += (int64_t)gridDim.x * 256) {
    const uint32_t u = e < total ? __float_as_uint(score[e]) : 0u;
    const bool in = e < total && (hi_shift >= 32 || (u >> hi_shift) == (prefix >> hi_shift));
    hist_add(sh, (u >> shift) & ((1u << bits) - 1u), in);
  }
  __syncthreads();
  uint32_t* dst = hist + pass * 2048;
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (sh[i]) atomicAdd(&dst[i], sh[i]);
}

constexpr int CAND_CAP = 4096;   // candidates the dense path gathers (= the sort buffer of select_emit_kernel)
constexpr int CAND_BUF = 32768;  // keys a pair's candidate buffer holds (fast path: every slab appends its list)

// gather every element with score >= threshold (bit pattern in hdr->prefix) as a sortable key
__global__ __launch_bounds__(256) void select_gather_kernel(const float* __restrict__ score,
                                                            SpmHdr* __restrict__ hdr,
                                                            const uint32_t* __restrict__ hist,
                                                            unsigned long long* __restrict__ cand, size_t zstride) {
  __shared__ uint32_t sh[2048];
  {
    const size_t zo = (size_t)blockIdx.z * zstride;
    score = z_shift(score, zo), hdr = z_shift(hdr, zo), hist = z_shift(hist, zo), cand = z_shift(cand, zo);
  }
  uint32_t prefix = 0;
  int k_rem = hdr->k;
  for (int pp = 0; pp < 3; ++pp) {
    const Pick pk = pick_from_hist(hist + pp * 2048, pp, prefix, k_rem, sh);
    prefix = pk.prefix;
    k_rem = pk.k_rem;
    __syncthreads();
  }
  const int ns = hdr->ns;
  const int64_t total = (int64_t)hdr->nr * ns;
  const uint32_t thr = hdr->k > 0 ? prefix : 0xffffffffu;  // exact bit pattern of the k-th largest score
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr->prefix = thr;  // (select_gather_ordered_kernel takes it from here)
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const uint32_t u = __float_as_uint(score[e]);
    if (u >= thr) {
      const int slot = atomicAdd(&hdr->n_cand, 1);
      // larger key = higher score, then LOWER flat index
      if (slot < CAND_CAP) cand[slot] = ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)e);
    }
  }
}

// Degenerate inputs (constant or all-zero features, padded superpoints passed without masks): more scores tie at the
// threshold than the candidate buffer holds.  torch.topk still returns k entries there, so does this path: ONE workgroup
// walks the score matrix in flat-index order, takes every score above the threshold (fewer than k of them) and the first
// k_rem scores equal to it -- exactly the set the sort would have kept (ties: lowest flat index first).
__global__ __launch_bounds__(1024) void select_gather_ordered_kernel(const float* __restrict__ score,
                                                                     SpmHdr* __restrict__ hdr,
                                                                     unsigned long long* __restrict__ cand) {
  __shared__ int s_w[2][1024 / WAVE];
  __shared__ int s_base[2];
  const int64_t total = (int64_t)hdr->nr * hdr->ns;
  const uint32_t thr = hdr->prefix;
  const int k = hdr->k;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if (threadIdx.x == 0) s_base[0] = s_base[1] = 0;  // [0] scores above the threshold so far, [1] ties so far
  __syncthreads();
  // how many ties may be kept = k - (#scores above the threshold), known only after a full sweep: first sweep counts
  int n_gt_local = 0;
  for (int64_t e = threadIdx.x; e < total; e += 1024) n_gt_local += __float_as_uint(score[e]) > thr ? 1 : 0;
  n_gt_local = wave_sum_i32_dpp(n_gt_local);
  if (lane == 0) s_w[0][wv] = n_gt_local;
  __syncthreads();
  int n_gt = 0;
  for (int i = 0; i < 1024 / WAVE; ++i) n_gt += s_w[0][i];
  const int tie_budget = k - n_gt;
  __syncthreads();
  for (int64_t e0 = 0; e0 < total; e0 += 1024) {
    const int64_t e = e0 + threadIdx.x;
    const uint32_t u = e < total ? __float_as_uint(score[e]) : 0u;
    const int gt = (e < total && u > thr) ? 1 : 0, eq = (e < total && u == thr) ? 1 : 0;
    const int igt = wave_incl_scan_add_dpp(gt), ieq = wave_incl_scan_add_dpp(eq);
    if (lane == WAVE - 1) {
      s_w[0][wv] = igt;
      s_w[1][wv] = ieq;
    }
    __syncthreads();
    int bgt = s_base[0], beq = s_base[1];
    for (int i = 0; i < wv; ++i) {
      bgt += s_w[0][i];
      beq += s_w[1][i];
    }
    const unsigned long long key = ((unsigned long long)u << 32) | (uint32_t)(~(uint32_t)e);
    if (gt) cand[bgt + igt - 1] = key;                       // slots [0, n_gt)
    const int tie_rank = beq + ieq - 1;
    if (eq && tie_rank < tie_budget) cand[n_gt + tie_rank] = key;  // slots [n_gt, k)
    __syncthreads();
    if (threadIdx.x == 1023) {
      s_base[0] = bgt + igt;
      s_base[1] = beq + ieq;
    }
    __syncthreads();
    if (s_base[1] >= tie_budget && s_base[0] >= n_gt) break;  // everything that will be kept has been seen
  }
  if (threadIdx.x == 0) hdr->n_cand = k;
}

// single block: sort the candidates (descending), emit the first k.  Candidates whose score lies below hdr->tau_max (fast
// path: a bound every one of the pair's k best scores reaches) are dropped first; what is left must fit the sort buffer
constexpr int EMIT_CAP = 4096;
__global__ __launch_bounds__(1024) void select_emit_kernel(SpmHdr* __restrict__ hdr,
                                                           const unsigned long long* __restrict__ cand,
                                                           const int32_t* __restrict__ ridx,
                                                           const int32_t* __restrict__ sidx,
                                                           int64_t* __restrict__ out_ref,
                                                           int64_t* __restrict__ out_src,
                                                           float* __restrict__ out_score, size_t zstride, int out_stride,
                                                           int cand_cap, int4* __restrict__ summary) {
  __shared__ unsigned long long sk[EMIT_CAP];
  __shared__ int s_n;
  {  // stack mode: workgroup = pair, its outputs are row blockIdx.x of the (pairs, num_correspondences) arrays
    const size_t zo = (size_t)blockIdx.x * zstride;
    hdr = z_shift(hdr, zo), cand = z_shift(cand, zo), ridx = z_shift(ridx, zo), sidx = z_shift(sidx, zo);
    out_ref += (int64_t)blockIdx.x * out_stride, out_src += (int64_t)blockIdx.x * out_stride;
    out_score += (int64_t)blockIdx.x * out_stride;
  }
  // what the host needs of every pair's header, in ONE contiguous row per pair (the headers themselves lie a workspace slice
  // apart: a strided read-back)
  auto report = [&](int overflow) {
    if (summary != nullptr && threadIdx.x == 0) summary[blockIdx.x] = make_int4(hdr->k, overflow, hdr->n_cand, 0);
  };
  if (hdr->overflow) {  // (uniform) the caller repeats this pair on the dense path
    report(1);
    return;
  }
  const int n_all = min(hdr->n_cand, cand_cap);
  const int k_sel = hdr->k;
  // Two bounds below the pair's k-th best score: hdr->tau_max (the slabs' thresholds), and -- over the candidates that reach
  // it -- the smallest of the 16 waves' ceil(k / 16)-th largest thread maxima (16 x ceil(k / 16) >= k candidates lie at or
  // above it).  Only candidates at or above both are sorted.
  constexpr int PER = CAND_BUF / 1024;  // candidates per thread, at most
  __shared__ unsigned int s_off[1024 / WAVE];
  unsigned int floor_u = hdr->tau_max;
  unsigned int uu[PER];
  unsigned int umax = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + j * 1024;
    uu[j] = i < n_all ? (unsigned int)(cand[i] >> 32) : 0u;
    if (i < n_all && uu[j] >= floor_u) umax = max(umax, uu[j]);
  }
  {
    unsigned int x = umax;
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int size = 2; size <= WAVE; size <<= 1) {
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        const unsigned int y = (unsigned int)__shfl_xor((int)x, stride, WAVE);
        const bool lower = (lane & stride) == 0, desc = (lane & size) == 0;
        x = (lower == desc) ? max(x, y) : min(x, y);
      }
    }
    const int want = (k_sel + 1024 / WAVE - 1) / (1024 / WAVE);
    const unsigned int offer = want >= 1 && want <= WAVE ? (unsigned int)__shfl((int)x, want - 1, WAVE) : 0u;
    if (lane == 0) s_off[threadIdx.x / WAVE] = offer;
  }
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  {
    unsigned int t2 = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < 1024 / WAVE; ++w) t2 = min(t2, s_off[w]);
    floor_u = max(floor_u, t2);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + j * 1024;
    if (i < n_all && uu[j] >= floor_u) {
      const int pos = atomicAdd(&s_n, 1);
      if (pos < EMIT_CAP) sk[pos] = cand[i];
    }
  }
  __syncthreads();
  const int n = s_n;
  if (n > EMIT_CAP || n < min(k_sel, n_all)) {  // uniform.  (Fewer than k survivors cannot happen while the bounds hold: checked anyway)
    if (threadIdx.x == 0) hdr->overflow = 1;
    report(1);
    return;
  }
  report(0);
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  for (int i = n + threadIdx.x; i < np2; i += 1024) sk[i] = 0ull;
  __syncthreads();
  if (np2 <= 1024) {
    // one key per thread, in a register: exchanges with a partner less than 64 positions away are lane shuffles (no LDS, no
    // barrier); only the few steps with a stride of 64 or more go through LDS -- 10 barrier pairs for 1 024 keys instead of
    // the 55 of the all-LDS network below (14 of this kernel's 18 us at a few hundred candidates)
    const int t = threadIdx.x;
    unsigned long long x = t < np2 ? sk[t] : 0ull;
    for (int size = 2; size <= np2; size <<= 1) {
      const bool desc = (t & size) == 0;
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        unsigned long long y;
        if (stride >= WAVE) {
          __syncthreads();
          sk[t] = x;
          __syncthreads();
          y = sk[t ^ stride];
        } else {
          const unsigned int lo32 = (unsigned int)__shfl_xor((int)(unsigned int)x, stride, WAVE);
          const unsigned int hi32 = (unsigned int)__shfl_xor((int)(unsigned int)(x >> 32), stride, WAVE);
          y = ((unsigned long long)hi32 << 32) | lo32;
        }
        const bool lower = (t & stride) == 0;
        x = (lower == desc) ? (x > y ? x : y) : (x < y ? x : y);
      }
    }
    __syncthreads();
    sk[t] = x;
    __syncthreads();
  } else
  // bitonic sort, descending, over the smallest power of two that holds the candidates
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < np2 / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = sk[lo], b = sk[hi];
        if ((a < b) == desc) {
          sk[lo] = b;
          sk[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  const int k = min(hdr->k, n), ns = hdr->ns;
  for (int i = threadIdx.x; i < k; i += 1024) {
    const unsigned long long key = sk[i];
    const uint32_t e = ~(uint32_t)(key & 0xffffffffull);
    out_ref[i] = ridx[e / (uint32_t)ns];   // superpoint_matching.py:44,47
    out_src[i] = sidx[e % (uint32_t)ns];   // :45,48
    out_score[i] = __uint_as_float((uint32_t)(key >> 32));
  }
}

// ---------------------------------------------------------------- selection, fast path
// One launch turns the (nr, ns) matrix of exp(-d) values into a short candidate list per pair -- instead of a dense score
// matrix written, three histogram sweeps over it and a gather sweep (five launches, ~6 passes over the matrix).  One
// workgroup of 1024 threads per SLAB of 48 matrix rows, thread = column: its 48 scores live in registers (priced with
// reciprocals first; the exact (s / rowsum) * (s / colsum) of normalize_kernel -- same bits -- for the listed entries only).
//   1. tau: every wave that holds 64 valid columns sorts its 64 thread maxima (bitonic network over the lanes) and offers
//      its ceil(k / full waves)-th largest; the smallest offer is a score that at least k scores of THIS SLAB reach, so it
//      is a lower bound of the pair's k-th best score;
//   2. every score >= tau goes to an LDS list as a key (score bits << 32 | ~flat index) -- a few hundred per slab -- and the
//      list is appended to the pair's candidates; the largest tau of the pair's slabs is kept in the header;
//   3. select_emit_kernel drops the candidates below that largest tau (each slab's tau bounds the pair's k-th best score
//      from below, so does their maximum), sorts the few hundred that are left and emits the first k.
// A slab list, candidate buffer or sort buffer that overflows (thousands of equal scores: degenerate inputs) raises
// hdr->overflow: the caller then takes the dense path below, which handles ties at any multiplicity.
constexpr int SS_T = 1024;       // threads per slab workgroup = columns it can hold
constexpr int SS_ROWS = 48;      // rows per slab = scores per thread (registers: 1024 threads leave 128 VGPRs each)
constexpr int SS_LIST = 2048;    // keys the slab's list holds
constexpr int SS_CODES = 4096;   // entries whose exact score is computed (listed by their approximate score)

__global__ __launch_bounds__(SS_T) void slab_select_kernel(const float* __restrict__ S, int ld, const float* __restrict__ rs,
                                                           const float* __restrict__ cs, int dual, SpmHdr* __restrict__ hdr,
                                                           unsigned long long* __restrict__ cand, int cand_cap,
                                                           const SpmStack* __restrict__ stack, size_t zstride) {
  __shared__ unsigned long long s_key[SS_LIST];
  __shared__ unsigned short s_code[SS_CODES];
  __shared__ float s_rr[SS_ROWS];
  __shared__ unsigned int s_tau[SS_T / WAVE];
  __shared__ int s_cnt, s_ncode, s_base;
  if (stack) {
    ld = stack[blockIdx.z].ns;
    const size_t zo = (size_t)blockIdx.z * zstride;
    S = z_shift(S, zo), rs = z_shift(rs, zo), cs = z_shift(cs, zo), hdr = z_shift(hdr, zo), cand = z_shift(cand, zo);
  }
  const int nr = hdr->nr, ns = hdr->ns, k = hdr->k;
  const int r0 = (int)blockIdx.x * SS_ROWS;
  if (r0 >= nr || k <= 0) return;  // (uniform: nobody waits at a barrier)
  const int rows = min(SS_ROWS, nr - r0);
  const int c = threadIdx.x;
  const bool col_ok = c < ns;
  const int cc = col_ok ? c : 0;
  // The exact score (two IEEE divisions per entry: 22 instructions) is only needed for the few hundred entries that can be
  // among the k best.  Everything is first priced with reciprocals: a = (s * rcp(rowsum)) * (s * rcp(colsum)), within
  // EPS = 2e-6 relative of the exact value (v_rcp_f32 is good to 1 ulp, three roundings follow: < 5e-7 in all).  A bound
  // tau_a that k approximate scores reach gives the exact bound tau = tau_a (1 - 2 EPS); entries whose approximate score
  // reaches tau_a (1 - 4 EPS) -- a superset of the entries whose exact score reaches tau -- are listed, and the exact score
  // is computed for the list only.  (An approximate score that is not a number lists its entry.)
  constexpr float EPS = 2.0e-6f;
  if (threadIdx.x < SS_ROWS) s_rr[threadIdx.x] = dual ? __builtin_amdgcn_rcpf(rs[r0 + min((int)threadIdx.x, rows - 1)]) : 1.0f;
  if (threadIdx.x == 0) s_cnt = 0, s_ncode = 0;
  const float rcs = dual ? __builtin_amdgcn_rcpf(cs[cc]) : 1.0f;
  __syncthreads();
  unsigned int u[SS_ROWS];  // approximate scores as bit patterns (>= 0: the patterns order like the values)
  unsigned int umax = 0u;
#pragma unroll
  for (int g = 0; g < SS_ROWS; g += 16) {
    float sv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sv[i] = S[(int64_t)(r0 + min(g + i, rows - 1)) * ld + cc];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float ap = dual ? (sv[i] * s_rr[g + i]) * (sv[i] * rcs) : sv[i];
      u[g + i] = __float_as_uint(ap);
      if (col_ok && g + i < rows) umax = max(umax, u[g + i]);  // (a NaN pattern sorts above every number)
    }
  }
  // ---- 1. a threshold that at least k (approximate) scores of the slab reach: every wave sorts its 64 thread maxima
  //      (bitonic network over the lanes, no barrier) and offers its ceil(k / full waves)-th largest; the smallest offer is
  //      the threshold
  {
    unsigned int x = col_ok ? umax : 0u;
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int size = 2; size <= WAVE; size <<= 1) {
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        const unsigned int y = (unsigned int)__shfl_xor((int)x, stride, WAVE);
        const bool lower = (lane & stride) == 0, desc = (lane & size) == 0;
        x = (lower == desc) ? max(x, y) : min(x, y);  // descending runs where (lane & size) == 0
      }
    }
    // only waves whose 64 columns all exist take part (the others offer "no bound"); without a full wave, or when the
    // full waves cannot supply k maxima, the threshold is 0: every score of the slab is a candidate
    const int nfull = ns / WAVE;
    const int want = nfull > 0 ? (k + nfull - 1) / nfull : WAVE + 1;
    unsigned int offer = 0xffffffffu;
    if (want <= WAVE) {
      const unsigned int kth = (unsigned int)__shfl((int)x, want - 1, WAVE);
      if ((int)(threadIdx.x / WAVE) < nfull) offer = kth;
    } else {
      offer = 0u;
    }
    if (lane == 0) s_tau[threadIdx.x / WAVE] = offer;
  }
  __syncthreads();
  unsigned int tau_a = 0xffffffffu;
#pragma unroll
  for (int w = 0; w < SS_T / WAVE; ++w) tau_a = min(tau_a, s_tau[w]);
  // the bounds in the exact domain / for the listing (a threshold that is not a finite number bounds nothing)
  const float taf = __uint_as_float(tau_a);
  const bool tau_ok = taf == taf && taf < INFINITY;
  const unsigned int tau = tau_ok ? __float_as_uint(taf * (1.0f - 2.0f * EPS)) : 0u;
  const float list_from = tau_ok ? taf * (1.0f - 4.0f * EPS) : 0.0f;
  // ---- 2a. list the entries that may reach tau: (row in slab, column) codes
  if (col_ok) {
#pragma unroll
    for (int i = 0; i < SS_ROWS; ++i) {
      if (i < rows && !(__uint_as_float(u[i]) < list_from)) {
        const int pos = atomicAdd(&s_ncode, 1);
        if (pos < SS_CODES) s_code[pos] = (unsigned short)(i * SS_T + c);
      }
    }
  }
  __syncthreads();
  const int ncode = s_ncode;
  // ---- 2b. the exact score of the listed entries (superpoint_matching.py:38-41: (s / rowsum) * (s / colsum)); those that reach
  //      tau are candidates of the pair
  for (int q = threadIdx.x; q < min(ncode, SS_CODES); q += SS_T) {
    const int code = s_code[q], i = code / SS_T, cq = code % SS_T;
    const float sq = S[(int64_t)(r0 + i) * ld + cq];
    const float ex = dual ? (sq / rs[r0 + i]) * (sq / cs[cq]) : sq;
    const unsigned int ue = __float_as_uint(ex);
    if (ue >= tau) {
      const int pos = atomicAdd(&s_cnt, 1);
      const unsigned int e = (unsigned int)(r0 + i) * (unsigned int)ns + (unsigned int)cq;  // flat index in the (nr, ns) matrix
      if (pos < SS_LIST) s_key[pos] = ((unsigned long long)ue << 32) | (unsigned int)(~e);
    }
  }
  __syncthreads();
  const int n = s_cnt;
  if (threadIdx.x == 0) {
    int base = 0;
    if (n <= SS_LIST && ncode <= SS_CODES) base = atomicAdd(&hdr->n_cand, n);
    if (n > SS_LIST || ncode > SS_CODES || base + n > cand_cap) {
      hdr->overflow = 1;
      base = -1;
    } else {
      atomicMax(&hdr->tau_max, tau);
    }
    s_base = base;
  }
  __syncthreads();
  const int base = s_base;
  if (base < 0) return;
  for (int i = threadIdx.x; i < n; i += SS_T) cand[base + i] = s_key[i];
}

// can the fast path take this shape?  (columns <= threads of a slab workgroup)
inline bool spm_fast_ok(int64_t nr, int64_t ns, int k) {
  (void)nr;
  return ns <= SS_T && k <= SS_T;
}

struct SpmWs {
  SpmHdr* hdr;
  int32_t* ridx;
  int32_t* sidx;
  float* S;
  float* score;
  float* rs;
  float* cs;
  float* rs_part;  // [ceil(ns / 64)][nr] partial row sums per tile column (distance kernel epilogue)
  float* cs_part;  // [ceil(nr / 64)][ns]
  uint32_t* hist;
  unsigned long long* cand;
  size_t bytes;
};

SpmWs carve_spm(void* p, int64_t nr, int64_t ns) {
  SpmWs w;
  Carver c(p);
  w.hdr = c.take<SpmHdr>(1);
  w.ridx = c.take<int32_t>(nr);
  w.sidx = c.take<int32_t>(ns);
  w.S = c.take<float>(nr * ns);
  w.score = c.take<float>(nr * ns);
  w.rs = c.take<float>(nr);
  w.cs = c.take<float>(ns);
  w.rs_part = c.take<float>(((ns + PD_T - 1) / PD_T) * nr);
  w.cs_part = c.take<float>(((nr + PD_T - 1) / PD_T) * ns);
  w.hist = c.take<uint32_t>(3 * 2048);
  w.cand = c.take<unsigned long long>(CAND_BUF);
  w.bytes = c.used();
  return w;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_pairwise_distance_batch_workspace_bytes(int64_t batch, int64_t n, int64_t m) {
  if (n < 0 || m < 0 || batch < 0) return 0;
  return align_up((size_t)std::max<int64_t>(batch, 1) * (align_up((size_t)n, 64) + align_up((size_t)m, 64)) * sizeof(float) + 512, 256);
}

// gr_pairwise_distance forwards to the batch entry with batch = 1: one size rule for both
extern "C" size_t gr_pairwise_distance_workspace_bytes(int64_t n, int64_t m) {
  return gr_pairwise_distance_batch_workspace_bytes(1, n, m);
}

namespace gr {
namespace {
// tile choice: 128 x 128 per workgroup once that still leaves every CU a workgroup or so; else 64 x 64
inline bool pd_use_big(int64_t n, int64_t m, int64_t batch) {
  return ((n + PB_T - 1) / PB_T) * ((m + PB_T - 1) / PB_T) * batch >= 192;
}
}  // namespace
}  // namespace gr

extern "C" int gr_pairwise_distance_batch(const float* x, const float* y, int64_t batch, int64_t n, int64_t m, int64_t c,
                                          int normalized, float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(batch >= 0 && n >= 0 && m >= 0 && c >= 0 && n < (1ll << 31) && m < (1ll << 31) && c < (1ll << 31) && batch < 65536 &&
                 batch * std::max(n, m) < (1ll << 31),
             "bad sizes");
  if (n == 0 || m == 0 || batch == 0) return GR_OK;
  GR_REQUIRE(x && y && out, "null argument");
  float* x2 = nullptr;
  float* y2 = nullptr;
  PdBatch bs{n * c, m * c, n * m, 0, 0};
  if (!normalized) {
    if (!ws || ws_bytes < gr_pairwise_distance_batch_workspace_bytes(batch, n, m)) {
      set_error("pairwise_distance workspace too small");
      return GR_ERR_WORKSPACE;
    }
    x2 = static_cast<float*>(ws);
    y2 = x2 + align_up((size_t)(batch * n), 64);
    bs.x2 = n;
    bs.y2 = m;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((batch * n + 3) / 4)), dim3(256), 0, stream, x, (int)(batch * n), (int)c, x2);
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((batch * m + 3) / 4)), dim3(256), 0, stream, y, (int)(batch * m), (int)c, y2);
  }
  KernelTimer timer("pairwise_distance", stream);
  if (pd_use_big(n, m, batch)) {
    const dim3 grid((unsigned)((m + PB_T - 1) / PB_T), (unsigned)((n + PB_T - 1) / PB_T), (unsigned)batch);
    const bool aligned = c % PB_K == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0;
    auto kern = aligned ? pairwise_big_kernel<EPI_DIST, true> : pairwise_big_kernel<EPI_DIST, false>;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, x, y, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (int)n, (int)m, (const int32_t*)nullptr, (int)c, normalized, x2, y2, out,
                       (int)m, (const SpmStack*)nullptr, (size_t)0, bs, (float*)nullptr, (float*)nullptr, 0, 0);
  } else {
    const dim3 grid((unsigned)((m + PD_T - 1) / PD_T), (unsigned)((n + PD_T - 1) / PD_T), (unsigned)batch);
    hipLaunchKernelGGL((pairwise_kernel<EPI_DIST>), grid, dim3(256), 0, stream, x, y, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (int)n, (int)m, (const int32_t*)nullptr, (int)c, normalized, x2, y2, out,
                       (int)m, (const SpmStack*)nullptr, (size_t)0, bs, (float*)nullptr, (float*)nullptr, 0, 0);
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

extern "C" int gr_pairwise_distance(const float* x, const float* y, int64_t n, int64_t m, int64_t c,
                                    int normalized, float* out, void* ws, size_t ws_bytes, void* stream_) {
  return gr_pairwise_distance_batch(x, y, 1, n, m, c, normalized, out, ws, ws_bytes, stream_);
}

extern "C" size_t gr_superpoint_matching_workspace_bytes(int64_t nr, int64_t ns) {
  if (nr < 0 || ns < 0) return 0;
  return carve_spm(nullptr, nr, ns).bytes;
}

// the dense selection: normalised score matrix, 3-digit radix select of the k-th largest score, gather, sort.  Shapes the slab
// kernel does not take, and pairs whose slabs overflowed (w.hist must be zero, w.rs / w.cs and hdr->k in place).
static void spm_select_dense(int64_t nr, int64_t ns, int num_correspondences, int dual_normalization, int64_t* out_ref_idx,
                             int64_t* out_src_idx, float* out_scores, const SpmWs& w, hipStream_t stream,
                             const SpmStack* stack, size_t zstride, int npairs, int4* summary = nullptr) {
  const unsigned z = (unsigned)npairs;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((ns + 255) / 256), (unsigned)nr, z), dim3(256), 0, stream, w.S,
                     (int)ns, w.rs, w.cs, dual_normalization, w.hdr, w.score, stack, zstride);
  const int sel_blocks = (int)std::min<int64_t>(stack ? 64 : 512, (nr * ns + 1023) / 1024);
  for (int pass = 0; pass < 3; ++pass)
    hipLaunchKernelGGL(select_hist_kernel, dim3(sel_blocks, 1, z), dim3(256), 0, stream, w.score, w.hdr, pass, w.hist, zstride);
  hipLaunchKernelGGL(select_gather_kernel, dim3(sel_blocks, 1, z), dim3(256), 0, stream, w.score, w.hdr, w.hist, w.cand,
                     zstride);
  hipLaunchKernelGGL(select_emit_kernel, dim3(z), dim3(1024), 0, stream, w.hdr, w.cand, w.ridx, w.sidx, out_ref_idx,
                     out_src_idx, out_scores, zstride, num_correspondences, CAND_CAP, summary);
}

// the launches of one pair -- or, with `stack`, of `npairs` pairs at once (nr / ns are then the largest counts, feats / masks
// the stacked arrays, w the first pair's workspace and zstride the distance to the next) -- asynchronous; the header
// (counts) stays in w.hdr
static int spm_launch(const float* ref_feats, const float* src_feats, int64_t nr, int64_t ns, int64_t c,
                      const uint8_t* ref_masks, const uint8_t* src_masks, int num_correspondences,
                      int dual_normalization, int64_t* out_ref_idx, int64_t* out_src_idx, float* out_scores,
                      const SpmWs& w, hipStream_t stream, const SpmStack* stack = nullptr, size_t zstride = 0,
                      int npairs = 1, int4* summary = nullptr) {
  const unsigned z = (unsigned)npairs;
  if (!stack && !spm_fast_ok(nr, ns, num_correspondences)) GR_HIP(hipMemsetAsync(w.hist, 0, 3 * 2048 * sizeof(uint32_t), stream));
  hipLaunchKernelGGL(compact_masks_kernel, dim3(z), dim3(1024), 0, stream, ref_masks, (int)nr, src_masks, (int)ns,
                     num_correspondences, w.ridx, w.sidx, w.hdr, w.hist, stack, zstride);
  // features are L2-normalised by the caller (model.py:143-144): d = 2 - 2 xy  (superpoint_matching.py:37)
  {
    KernelTimer timer("spm_distance", stream);
    const PdBatch none{0, 0, 0, 0, 0};
    if (pd_use_big(nr, ns, npairs)) {
      const dim3 grid((unsigned)((ns + PB_T - 1) / PB_T), (unsigned)((nr + PB_T - 1) / PB_T), z);
      const bool aligned = c % PB_K == 0 && (reinterpret_cast<uintptr_t>(ref_feats) | reinterpret_cast<uintptr_t>(src_feats)) % 16 == 0;
      auto kern = aligned ? pairwise_big_kernel<EPI_EXPNEG, true> : pairwise_big_kernel<EPI_EXPNEG, false>;
      hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, ref_feats, src_feats, w.ridx, w.sidx,
                         (int)nr, (int)ns, reinterpret_cast<const int32_t*>(w.hdr), (int)c, 1, (const float*)nullptr,
                         (const float*)nullptr, w.S, (int)ns, stack, zstride, none, dual_normalization ? w.rs_part : nullptr,
                         dual_normalization ? w.cs_part : nullptr, (int)nr, (int)ns);
    } else {
      const dim3 grid((unsigned)((ns + PD_T - 1) / PD_T), (unsigned)((nr + PD_T - 1) / PD_T), z);
      hipLaunchKernelGGL((pairwise_kernel<EPI_EXPNEG>), grid, dim3(256), 0, stream, ref_feats, src_feats, w.ridx, w.sidx,
                         (int)nr, (int)ns, reinterpret_cast<const int32_t*>(w.hdr), (int)c, 1, (const float*)nullptr,
                         (const float*)nullptr, w.S, (int)ns, stack, zstride, none, dual_normalization ? w.rs_part : nullptr,
                         dual_normalization ? w.cs_part : nullptr, (int)nr, (int)ns);
    }
  }
  if (dual_normalization) {
    const int tile = pd_use_big(nr, ns, npairs) ? PB_T : PD_T;  // (the kernel that ran above)
    hipLaunchKernelGGL(sums_finish_kernel, dim3((unsigned)((std::max(nr, ns) + 255) / 256), 1, z), dim3(256), 0, stream, w.hdr,
                       w.rs_part, w.cs_part, (int)nr, (int)ns, tile, w.rs, w.cs, stack, zstride);
  }
  if (spm_fast_ok(nr, ns, num_correspondences)) {
    // the k best of every 64-row slab -> the pair's candidates (no dense score matrix, no histogram sweeps)
    hipLaunchKernelGGL(slab_select_kernel, dim3((unsigned)((nr + SS_ROWS - 1) / SS_ROWS), 1, z), dim3(SS_T), 0, stream, w.S,
                       (int)ns, w.rs, w.cs, dual_normalization, w.hdr, w.cand, CAND_BUF, stack, zstride);
    hipLaunchKernelGGL(select_emit_kernel, dim3(z), dim3(1024), 0, stream, w.hdr, w.cand, w.ridx, w.sidx, out_ref_idx,
                       out_src_idx, out_scores, zstride, num_correspondences, CAND_BUF, summary);
  } else {
    spm_select_dense(nr, ns, num_correspondences, dual_normalization, out_ref_idx, out_src_idx, out_scores, w, stream, stack,
                     zstride, npairs, summary);
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

static int spm_check_args(int64_t nr, int64_t ns, int64_t c, int num_correspondences) {
  GR_REQUIRE(nr >= 0 && ns >= 0 && c >= 0 && num_correspondences >= 0, "bad sizes");
  GR_REQUIRE(nr * ns < (1ll << 32), "score matrix too large (%lld x %lld)", (long long)nr, (long long)ns);
  GR_REQUIRE(num_correspondences <= CAND_CAP / 2, "num_correspondences must be <= %d", CAND_CAP / 2);
  return GR_OK;
}

extern "C" int gr_superpoint_matching(const float* ref_feats, const float* src_feats, int64_t nr, int64_t ns,
                                      int64_t c, const uint8_t* ref_masks, const uint8_t* src_masks,
                                      int num_correspondences, int dual_normalization, int64_t* out_ref_idx,
                                      int64_t* out_src_idx, float* out_scores, int64_t* h_num_out, void* ws,
                                      size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_num_out != nullptr, "h_num_out is null");
  *h_num_out = 0;
  int rc = spm_check_args(nr, ns, c, num_correspondences);
  if (rc != GR_OK) return rc;
  if (nr == 0 || ns == 0 || num_correspondences == 0) return GR_OK;
  GR_REQUIRE(ref_feats && src_feats && out_ref_idx && out_src_idx && out_scores, "null argument");
  SpmWs w = carve_spm(ws, nr, ns);
  if (!ws || ws_bytes < w.bytes) {
    set_error("superpoint_matching workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return GR_ERR_WORKSPACE;
  }
  rc = spm_launch(ref_feats, src_feats, nr, ns, c, ref_masks, src_masks, num_correspondences, dual_normalization,
                  out_ref_idx, out_src_idx, out_scores, w, stream);
  if (rc != GR_OK) return rc;
  // (the header comes back through pinned memory: a copy into pageable memory is staged by the runtime, ~20 us of a 0.1 ms call)
  SpmHdr* hp = static_cast<SpmHdr*>(pinned_scratch(8, sizeof(SpmHdr)));
  GR_REQUIRE(hp != nullptr, "pinned read-back buffer could not be allocated");
  SpmHdr& h = *hp;
  GR_HIP(hipMemcpyAsync(&h, w.hdr, sizeof(h), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  bool dense = !spm_fast_ok(nr, ns, num_correspondences);
  if (!dense && h.overflow) {
    dense = true;
    // fast path: a slab list, the candidate buffer or the sort buffer overflowed (thousands of scores at a threshold): the
    // dense selection for this pair (counters and histograms cleared first)
    GR_HIP(hipMemsetAsync(&w.hdr->n_cand, 0, 3 * sizeof(int32_t), stream));  // n_cand, tau_max, overflow
    GR_HIP(hipMemsetAsync(w.hist, 0, 3 * 2048 * sizeof(uint32_t), stream));
    spm_select_dense(nr, ns, num_correspondences, dual_normalization, out_ref_idx, out_src_idx, out_scores, w, stream, nullptr, 0, 1);
    GR_LAUNCH_CHECK();
    GR_HIP(hipMemcpyAsync(&h, w.hdr, sizeof(h), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
  }
  if (dense && h.n_cand > CAND_CAP) {
    // more ties at the threshold than the candidate buffer holds: redo the gather in flat-index order (see above)
    hipLaunchKernelGGL(select_gather_ordered_kernel, dim3(1), dim3(1024), 0, stream, w.score, w.hdr, w.cand);
    hipLaunchKernelGGL(select_emit_kernel, dim3(1), dim3(1024), 0, stream, w.hdr, w.cand, w.ridx, w.sidx, out_ref_idx,
                       out_src_idx, out_scores, (size_t)0, 0, CAND_CAP, (int4*)nullptr);
    GR_LAUNCH_CHECK();
    GR_HIP(hipStreamSynchronize(stream));
  }
  *h_num_out = h.k;
  return GR_OK;
}

// Stack mode over `npairs` scene pairs (test.py:146-212 runs model.py:156-160 once per pair): the superpoint features of the
// batch are stacked as [ref_0, src_0, ref_1, src_1, ...] with h_node_off (2 npairs + 1 offsets), masks likewise (null =
// all true).  Pair b's matches go to row b of the (npairs, num_correspondences) outputs; h_num_out[b] = how many are valid.
// ONE set of five launches for the whole batch (grid.z = pair; every pair owns an equally laid out slice of the
// workspace, sized for the largest pair), the pairs' headers read back ONCE.  (A pair with more ties at the selection
// threshold than the candidate buffer holds is redone through the single-pair path afterwards.)
static void spm_max_sizes(const int64_t* h_node_off, int64_t npairs, int64_t* max_nr, int64_t* max_ns) {
  *max_nr = *max_ns = 0;
  for (int64_t b = 0; b < npairs; ++b) {
    *max_nr = std::max(*max_nr, h_node_off[2 * b + 1] - h_node_off[2 * b]);
    *max_ns = std::max(*max_ns, h_node_off[2 * b + 2] - h_node_off[2 * b + 1]);
  }
}

extern "C" size_t gr_superpoint_matching_batch_workspace_bytes(const int64_t* h_node_off, int64_t npairs) {
  if (!h_node_off || npairs < 0) return 0;
  int64_t mr, ms;
  spm_max_sizes(h_node_off, npairs, &mr, &ms);
  return (size_t)std::max<int64_t>(npairs, 1) * align_up(carve_spm(nullptr, mr, ms).bytes, 256) +
         align_up((size_t)std::max<int64_t>(npairs, 1) * sizeof(SpmStack), 256) +
         align_up((size_t)std::max<int64_t>(npairs, 1) * sizeof(int4), 256);
}

extern "C" int gr_superpoint_matching_batch(const float* feats, const int64_t* h_node_off, int64_t npairs, int64_t c,
                                            const uint8_t* masks, int num_correspondences, int dual_normalization,
                                            int64_t* out_ref_idx, int64_t* out_src_idx, float* out_scores,
                                            int64_t* h_num_out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(npairs >= 0 && npairs < 65536 && h_node_off && h_num_out, "bad arguments");
  for (int64_t b = 0; b < npairs; ++b) h_num_out[b] = 0;
  if (npairs == 0 || num_correspondences == 0) return GR_OK;
  GR_REQUIRE(feats && out_ref_idx && out_src_idx && out_scores, "null argument");
  int64_t mr, ms;
  spm_max_sizes(h_node_off, npairs, &mr, &ms);
  for (int64_t b = 0; b < npairs; ++b) {
    const int64_t nr = h_node_off[2 * b + 1] - h_node_off[2 * b], ns = h_node_off[2 * b + 2] - h_node_off[2 * b + 1];
    GR_REQUIRE(nr >= 0 && ns >= 0 && h_node_off[2 * b] >= 0, "pair %lld: offsets must ascend", (long long)b);
    GR_REQUIRE(h_node_off[2 * b + 2] < (1ll << 31), "too many superpoints");
  }
  int rc = spm_check_args(mr, ms, c, num_correspondences);
  if (rc != GR_OK) return rc;
  if (!ws || ws_bytes < gr_superpoint_matching_batch_workspace_bytes(h_node_off, npairs)) {
    set_error("superpoint_matching batch workspace too small");
    return GR_ERR_WORKSPACE;
  }
  if (mr == 0 || ms == 0) return GR_OK;
  const size_t pair_bytes = align_up(carve_spm(nullptr, mr, ms).bytes, 256);
  SpmStack* d_stack = reinterpret_cast<SpmStack*>(static_cast<char*>(ws) + (size_t)npairs * pair_bytes);
  // the pair table goes up through pinned per-thread staging; an event says when the copy has left it
  static thread_local hipEvent_t staged = nullptr;
  if (staged == nullptr) GR_HIP(hipEventCreateWithFlags(&staged, hipEventDisableTiming));
  else GR_HIP(hipEventSynchronize(staged));
  SpmStack* h_stack = static_cast<SpmStack*>(pinned_scratch(6, sizeof(SpmStack) * (size_t)npairs));
  GR_REQUIRE(h_stack != nullptr, "pinned staging buffer could not be allocated");
  for (int64_t b = 0; b < npairs; ++b) {
    h_stack[b].r0 = (int32_t)h_node_off[2 * b];
    h_stack[b].nr = (int32_t)(h_node_off[2 * b + 1] - h_node_off[2 * b]);
    h_stack[b].s0 = (int32_t)h_node_off[2 * b + 1];
    h_stack[b].ns = (int32_t)(h_node_off[2 * b + 2] - h_node_off[2 * b + 1]);
  }
  GR_HIP(hipMemcpyAsync(d_stack, h_stack, sizeof(SpmStack) * (size_t)npairs, hipMemcpyHostToDevice, stream));
  GR_HIP(hipEventRecord(staged, stream));
  SpmWs w = carve_spm(ws, mr, ms);  // pair 0's slice; pair z's is zstride = pair_bytes further
  int4* d_sum = reinterpret_cast<int4*>(reinterpret_cast<char*>(d_stack) + align_up((size_t)npairs * sizeof(SpmStack), 256));
  rc = spm_launch(feats, feats, mr, ms, c, masks, masks, num_correspondences, dual_normalization, out_ref_idx, out_src_idx,
                  out_scores, w, stream, d_stack, pair_bytes, (int)npairs, d_sum);
  if (rc != GR_OK) return rc;
  int4* hb = static_cast<int4*>(pinned_scratch(8, sizeof(int4) * (size_t)npairs));
  GR_REQUIRE(hb != nullptr, "pinned read-back buffer could not be allocated");
  GR_HIP(hipMemcpyAsync(hb, d_sum, sizeof(int4) * (size_t)npairs, hipMemcpyDeviceToHost, stream));  // (k, overflow, n_cand) per pair
  GR_HIP(hipStreamSynchronize(stream));
  const std::vector<int4> h(hb, hb + npairs);  // (a pair redone below goes through the single-pair entry, which reuses the slot)
  for (int64_t b = 0; b < npairs; ++b) {
    const int64_t r0 = h_node_off[2 * b], nr = h_node_off[2 * b + 1] - r0, s0 = h_node_off[2 * b + 1],
                  ns = h_node_off[2 * b + 2] - s0;
    if (nr == 0 || ns == 0) continue;  // (its header says k = 0 as well)
    if (spm_fast_ok(mr, ms, num_correspondences) ? h[b].y != 0 : h[b].z > CAND_CAP) {
      rc = gr_superpoint_matching(feats + r0 * c, feats + s0 * c, nr, ns, c, masks ? masks + r0 : nullptr,
                                  masks ? masks + s0 : nullptr, num_correspondences, dual_normalization,
                                  out_ref_idx + b * num_correspondences, out_src_idx + b * num_correspondences,
                                  out_scores + b * num_correspondences, h_num_out + b, ws, pair_bytes, stream_);
      if (rc != GR_OK) return rc;
    } else {
      h_num_out[b] = h[b].x;
    }
  }
  return GR_OK;
}
