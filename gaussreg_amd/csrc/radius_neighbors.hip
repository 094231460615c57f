// Fixed-radius neighbour search on MI355X: uniform-grid cell binning + per-query 27-cell scan.
//
// Replaces the reference's kd-tree path
//   geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
// with a design that has no tree at all: the result of the reference depends only on
//   (1) fp32 d = ((dx*dx + dy*dy) + dz*dz)      (nanoflann.hpp:432-440)
//   (2) strict d < r*r                          (nanoflann.hpp:249-253)
//   (3) ascending-d order per query             (nanoflann.hpp:1287)
// so any exhaustive candidate enumeration that applies (1)-(3) is result-identical.  This file is
// compiled with -ffp-contract=off so (1) stays three multiplies and two adds.
//
// Pipeline (all on `stream`):
//   bbox        per-cloud bounding box of the supports: per-block partial boxes, folded by grid_setup (no atomics)
//   grid_setup  per-cloud cell edge (>= radius, coarsened so cells <= max(4096, 4 n_b)), dims, cell and super-cell bases
//   binning     counting sort in two levels: points -> super-cells of 512 consecutive cells (block-local LDS histograms, one
//               global atomic per block and non-empty super-cell), then one workgroup per super-cell sorts its points by
//               cell in LDS and writes the cell table and the cell-ordered float4 {x,y,z,orig index} array
//   count       thread per (cell-ordered query, z-slab): candidates staged in LDS as coordinate planes, tested two at
//               a time with packed fp32 math; per thread a hit count, the nine candidate ranges and a hit bit mask;
//               max over queries -> host (the row width the reference returns)
//   fill        gathers only the hits named by the masks into per-query LDS segments, ranks every hit inside its
//               segment by counting (one thread per hit) and stores it at out[query][rank]; pads the rows
//   fused       (gr_radius_search mode 1) count + fill in ONE kernel for a width known before the launch
// gr_radius_count_cached lets consecutive searches over the same supports and radius skip bbox .. binning for the
// support side (the data pyramid searches every level's supports three times).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace gr {
namespace {

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, WAVE));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, WAVE));
  return v;
}

struct BatchGrid {  // 64 bytes: copied to LDS as four int4
  double org[3];
  double inv_cell;    // y and z: cells of edge >= r (1 + 2^-10)
  double inv_cell_x;  // x: `xk` sub-cells per cell -- the x window of a query shrinks from 3 r to (2 + 1/xk) r while every
                      // (y, z) row of cells stays one contiguous range of the cell-sorted supports (x is the fastest index)
  int dim[3];         // dim[0] counts the fine x cells
  int cell_base;
  int xk;
  int sup_base;  // first super-cell (SUP_CELLS consecutive cells) of this cloud
};
static_assert(sizeof(BatchGrid) == 64, "BatchGrid is staged in LDS as four int4");

struct RadiusHdr {
  unsigned int max_count;
  unsigned int max_block_hits;
  int total_cells;
  int total_sup;  // super-cells of all clouds
  int slow_sum;   // (thread-per-query kernel) queries the network could not finish: finished exactly by their wave
};

constexpr int RT = 128;
// binning: counting sort in two levels -- points -> super-cells of SUP_CELLS consecutive cells (block-local LDS histograms,
// one global atomic per block and non-empty super-cell), then one workgroup per super-cell sorts its points by cell in LDS
constexpr int SUP_SHIFT = 9, SUP_CELLS = 1 << SUP_SHIFT;
constexpr int COARSE_PTS = 2048;   // points per block of the coarse passes
constexpr int COARSE_BINS = 4096;  // super-cell range a block can histogram in LDS
constexpr int BBOX_PTS = 2048;     // points per block of the bounding-box pass  // queries per block in count/fill (3 threads per query)

struct RadiusWs {
  RadiusHdr* hdr;
  int32_t* q_off;
  int32_t* s_off;
  uint32_t* bbox;
  uint32_t* bbox_partial;  // [blocks][6]
  int32_t* blk_off;
  BatchGrid* grids;
  int32_t* sup_off;    // [batch+1] first super-cell of every cloud
  int32_t* sup_zero;   // [4][nsup+1]: counts (s, q) and cursors (s, q) -- cleared per call
  int32_t* sup_start;  // [2][nsup+1]
  int32_t* s_cell;
  int32_t* q_cell;
  int2* pairs_s;  // (point index, cell), grouped by super-cell
  int2* pairs_q;
  int32_t* start;  // [2][ccap+1] cell starts in the sorted arrays (supports; the query row is unused)
  float4* sorted_s;
  float4* sorted_q;
  int32_t* q_count;    // [3][nq] hits per (z-slab, query)
  int2* q_rng;         // [3 dy][3 slab][nq] candidate range (p0, p1) per band
  unsigned long long* q_mask;  // [3 slab][nq] hit bits in candidate enumeration order
  int32_t* blk_stats;  // [blocks][2]
  float* plane_x;      // [ns + 8] cell-ordered coordinate planes of the supports (tq_kernel)
  float* plane_y;
  float* plane_z;
  uint32_t* tiles;     // [nq rounded up to 64][TQ_ROW_CAP] sorted neighbour indices per query, cell order (tq_kernel, compact mode)
  int64_t ccap;
  int64_t nsup;  // upper bound of the number of super-cells
  size_t bytes;
};

RadiusWs carve(void* ws, int64_t nq, int64_t ns, int64_t batch) {
  RadiusWs w;
  Carver c(ws);
  w.ccap = 4096 * batch + 4 * ns;
  w.nsup = w.ccap / SUP_CELLS + batch + 2;
  w.hdr = c.take<RadiusHdr>(1);
  w.q_off = c.take<int32_t>(3 * (batch + 1));  // q offsets | s offsets | bbox block offsets: one host-to-device copy
  w.s_off = w.q_off + (batch + 1);
  w.blk_off = w.s_off + (batch + 1);
  w.bbox = c.take<uint32_t>(batch * 6);
  w.bbox_partial = c.take<uint32_t>(6 * (ns / BBOX_PTS + batch + 1));
  w.grids = c.take<BatchGrid>(batch);
  w.sup_off = c.take<int32_t>(batch + 1);
  // support side first (sizes depend on ns and batch only): a later call with other queries finds it in place
  w.sup_zero = c.take<int32_t>(4 * (w.nsup + 1));
  w.sup_start = c.take<int32_t>(2 * (w.nsup + 1));
  w.s_cell = c.take<int32_t>(ns);
  w.pairs_s = c.take<int2>(ns);
  w.start = c.take<int32_t>(2 * (w.ccap + 1));
  w.sorted_s = c.take<float4>(ns);
  w.plane_x = c.take<float>(ns + 8);
  w.plane_y = c.take<float>(ns + 8);
  w.plane_z = c.take<float>(ns + 8);
  // query side
  w.q_cell = c.take<int32_t>(nq);
  w.pairs_q = c.take<int2>(nq);
  w.sorted_q = c.take<float4>(nq);
  w.q_count = c.take<int32_t>(3 * nq);
  w.q_rng = c.take<int2>(9 * nq);
  w.q_mask = c.take<unsigned long long>(3 * nq);
  w.blk_stats = c.take<int32_t>(2 * ((nq + 63) / 64 + 8));  // fused_kernel runs 64 queries per workgroup
  w.tiles = c.take<uint32_t>((size_t)((nq + 63) / 64) * 64 * 64);
  w.bytes = c.used();
  return w;
}

// ---------------------------------------------------------------- grid setup
// Per-cloud bounding boxes without atomics: a block reduces one BBOX_PTS-point slice of ONE cloud into six words of
// `partial` (same-address global atomics cost ~60 ns each across XCDs: the shared bbox_kernel of common.hip, six atomics
// per 1024 points, took 32 us of the 8 x 200 k binning); grid_setup_kernel folds the partials.
// Offsets of a call with few clouds travel in the kernel arguments of the FIRST launch (q offsets | s offsets | bbox block
// offsets, nb + 1 entries each): no host -> device copy in front of the binning (a copy-engine operation and the hand-over to
// the first kernel: ~8 us of a 0.36 ms search); block 0 leaves them in the workspace for the launches behind.
constexpr int KARG_CLOUDS = 64;
struct OffsetArgs {
  int32_t v[3 * (KARG_CLOUDS + 1)];
};

template <bool KARG>
__global__ __launch_bounds__(256) void bbox_partial_kernel(const float* __restrict__ pts, const int32_t* __restrict__ off_dev,
                                                           const int32_t* __restrict__ blk_off_dev, int nb,
                                                           uint32_t* __restrict__ partial, int32_t* __restrict__ zero,
                                                           int nzero, const OffsetArgs ka, int32_t* __restrict__ q_off_out) {
  __shared__ uint32_t red[6][256 / WAVE];
  // (the super-cell counters of the counting sort are cleared here, by the way: one launch less in front of every search)
  for (int k = blockIdx.x * 256 + threadIdx.x; k < nzero; k += gridDim.x * 256) zero[k] = 0;
  const int32_t* off = KARG ? ka.v + (nb + 1) : off_dev;           // (the supports' offsets)
  const int32_t* blk_off = KARG ? ka.v + 2 * (nb + 1) : blk_off_dev;
  if (KARG && blockIdx.x == 0)
    for (int k = threadIdx.x; k < 3 * (nb + 1); k += 256) q_off_out[k] = ka.v[k];  // q_off | s_off | blk_off are neighbours
  const int b0 = find_batch(blk_off, nb, (int)blockIdx.x);
  const int p_first = off[b0] + ((int)blockIdx.x - blk_off[b0]) * BBOX_PTS;
  const int p_end = min(off[b0 + 1], p_first + BBOX_PTS);
  const int64_t f0 = (int64_t)p_first * 3;
  const int count = (p_end - p_first) * 3;
  // element f of the flat float stream belongs to axis (f0 + f) % 3; the stride 256 = 1 (mod 3), so a thread's
  // consecutive elements cycle through the axes: slot k of (l3, h3) holds axis (ax0 + k) % 3
  const int ax0 = (int)((f0 + threadIdx.x) % 3);
  const float* src = pts + f0;
  uint32_t l3[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, h3[3] = {0u, 0u, 0u};
  for (int f = threadIdx.x; f < count; f += 6 * 256) {
    uint32_t v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = f2ord(src[min(f + k * 256, count - 1)]);  // six independent loads in flight
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (f + k * 256 < count) {
        l3[k % 3] = min(l3[k % 3], v[k]);
        h3[k % 3] = max(h3[k % 3], v[k]);
      }
  }
  uint32_t lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int k = a - ax0 < 0 ? a - ax0 + 3 : a - ax0;  // slot that holds axis a
    lo[a] = k == 0 ? l3[0] : (k == 1 ? l3[1] : l3[2]);
    hi[a] = k == 0 ? h3[0] : (k == 1 ? h3[1] : h3[2]);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = (uint32_t)wave_min_u32(lo[a]);
    hi[a] = (uint32_t)wave_max_u32(hi[a]);
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[a][w] = lo[a];
      red[3 + a][w] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = red[threadIdx.x][0];
#pragma unroll
    for (int i = 1; i < 256 / WAVE; ++i) v = threadIdx.x < 3 ? min(v, red[threadIdx.x][i]) : max(v, red[threadIdx.x][i]);
    partial[(int64_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__global__ void grid_setup_kernel(uint32_t* __restrict__ bbox, const uint32_t* __restrict__ partial,
                                  const int32_t* __restrict__ blk_off,
                                  const int32_t* __restrict__ s_off, int nb, float radius, int xk_max,
                                  BatchGrid* __restrict__ grids, RadiusHdr* __restrict__ hdr,
                                  int32_t* __restrict__ sup_off) {
  // fold the per-block partial boxes: one wave per cloud (looped), lanes over the cloud's blocks
  {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    for (int b = w; b < nb; b += nw) {
      uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
      for (int k = blk_off[b] + lane; k < blk_off[b + 1]; k += WAVE) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lo[a] = min(lo[a], partial[(int64_t)k * 6 + a]);
          hi[a] = max(hi[a], partial[(int64_t)k * 6 + 3 + a]);
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = (uint32_t)wave_min_u32(lo[a]);
        hi[a] = (uint32_t)wave_max_u32(hi[a]);
      }
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          bbox[b * 6 + a] = lo[a];
          bbox[b * 6 + 3 + a] = hi[a];
        }
      }
    }
    __syncthreads();
  }
  // one thread per cloud (looped), then a serial prefix by thread 0 (nb is small)
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    BatchGrid g;
    const int n_b = s_off[b + 1] - s_off[b];
    double cell = fabs((double)radius) * (1.0 + 1.0 / 1024.0);
    if (!(cell > 0.0) || !isfinite(cell)) cell = 1.0;
    g.dim[0] = g.dim[1] = g.dim[2] = 1;
    g.org[0] = g.org[1] = g.org[2] = 0.0;
    if (n_b > 0) {
      double mn[3], mx[3];
      bool finite = true;
      for (int k = 0; k < 3; ++k) {
        mn[k] = (double)ord2f(bbox[b * 6 + k]);
        mx[k] = (double)ord2f(bbox[b * 6 + 3 + k]);
        finite = finite && isfinite(mn[k]) && isfinite(mx[k]);
        g.org[k] = mn[k];
      }
      if (finite) {
        const double cap = (double)max(4096, 4 * n_b);
        bool ok = false;
        for (int it = 0; it < 256; ++it) {
          double e[3], tot = 1.0;
          for (int k = 0; k < 3; ++k) {
            e[k] = floor((mx[k] - mn[k]) / cell) + 1.0;
            tot *= e[k];
          }
          if (tot <= cap) {
            for (int k = 0; k < 3; ++k) g.dim[k] = (int)e[k];
            ok = true;
            break;
          }
          cell *= fmax(cbrt(tot / cap), 1.05);
        }
        if (!ok) cell = INFINITY;  // one cell holds everything (inv_cell = 0): brute force
      } else {
        for (int k = 0; k < 3; ++k) g.org[k] = 0.0;
      }
    }
    g.inv_cell = 1.0 / cell;
    g.inv_cell_x = g.inv_cell;
    g.xk = 1;
    g.sup_base = 0;
    if (n_b > 0 && isfinite(cell) && isfinite(g.inv_cell) && g.inv_cell > 0.0) {
      // refine x only: a support within r of a query is within +-k fine cells of it (|dx| k / cell < k / (1 + 2^-10))
      const double cap = (double)max(4096, 4 * n_b);
      const double ext = (double)ord2f(bbox[b * 6 + 3]) - (double)ord2f(bbox[b * 6]);
      for (int k = xk_max; k > 1; k >>= 1) {
        const double inv_x = (double)k / cell;
        const double ex = floor(ext * inv_x) + 1.0;
        if (isfinite(ex) && ex * (double)g.dim[1] * (double)g.dim[2] <= cap && ex < 2147483647.0) {
          g.xk = k;
          g.inv_cell_x = inv_x;
          g.dim[0] = (int)ex;
          break;
        }
      }
    }
    g.cell_base = 0;
    g.sup_base = 0;
    grids[b] = g;
  }
  // exclusive prefix of the per-cloud cell counts: chunks of blockDim clouds, running carry in LDS
  // (a serial loop over global memory cost ~0.2 us per cloud)
  __shared__ int s_cnt[256], s_sup[256];
  __shared__ int s_carry, s_carry_sup;
  if (threadIdx.x == 0) s_carry = s_carry_sup = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
    const int b = b0 + threadIdx.x;
    const int cells = b < nb ? grids[b].dim[0] * grids[b].dim[1] * grids[b].dim[2] : 0;
    s_cnt[threadIdx.x] = cells;
    s_sup[threadIdx.x] = (cells + SUP_CELLS - 1) >> SUP_SHIFT;
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = s_carry, acs = s_carry_sup;
      const int live = min((int)blockDim.x, nb - b0);  // (a walk over all 256 slots for 8 clouds was 8 of this launch's 9.6 us)
      for (int k = 0; k < live; ++k) {
        const int c = s_cnt[k], u = s_sup[k];
        s_cnt[k] = acc;
        s_sup[k] = acs;
        acc += c;
        acs += u;
      }
      s_carry = acc;
      s_carry_sup = acs;
    }
    __syncthreads();
    if (b < nb) {
      grids[b].cell_base = s_cnt[threadIdx.x];
      grids[b].sup_base = s_sup[threadIdx.x];
      sup_off[b] = s_sup[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hdr->total_cells = s_carry;
    hdr->total_sup = s_carry_sup;
    hdr->max_count = 0;
    hdr->max_block_hits = 0;
    sup_off[nb] = s_carry_sup;
  }
}

__device__ inline double cell_coord(float x, double org, double inv) {
  return floor(((double)x - org) * inv);
}

__device__ inline int clamped_cell(const BatchGrid& g, float x, float y, float z) {
  int c[3];
  const float p[3] = {x, y, z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double u = cell_coord(p[k], g.org[k], k == 0 ? g.inv_cell_x : g.inv_cell);
    u = fmin(fmax(u, 0.0), (double)(g.dim[k] - 1));  // NaN -> 0
    c[k] = (int)u;
  }
  return g.cell_base + c[0] + g.dim[0] * (c[1] + g.dim[1] * c[2]);
}

// ---------------------------------------------------------------- binning: two-level counting sort
// (Round 1-2 counted with one returning global atomic per point: device-scope atomics are served behind the per-XCD L2s,
// 1.6 M of them took 64 us, plus a 25 MB clear of the cell table and a three-launch scan over it.)
struct BinSide {
  const float* pts;
  int n;
  const int32_t* off;     // [nb+1] cloud offsets
  int32_t* cell;          // [n] cell of every point (written by the counting pass, read by the scatter pass)
  int2* pairs;            // [n] (point, cell) grouped by super-cell
  int32_t* sup_cnt;       // [nsup+1]
  int32_t* sup_cur;       // [nsup+1]
  int32_t* sup_start;     // [nsup+1]
  float4* sorted;         // [n] {x, y, z, original index} in cell order
  int32_t* cell_start;    // [cells+1] or null (queries need no cell table)
  float* plane_x;         // [n + 8] or null: the same order as coordinate planes (supports only)
  float* plane_y;
  float* plane_z;
};

__global__ void bin_init_kernel(uint32_t* __restrict__ bbox, int nb, int32_t* __restrict__ zero, int nzero) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (bbox && i < nb * 6) bbox[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
  for (int k = i; k < nzero; k += gridDim.x * blockDim.x) zero[k] = 0;
}

__global__ void bin_init2_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int n) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) a[k] = 0, b[k] = 0;
}

// COUNT: cell of every point, LDS histogram over the block's super-cells, one global add per non-empty super-cell.
// SCATTER: the same histogram hands every point its rank inside (block, super-cell); one returning global add per
// non-empty super-cell reserves the block's span; (point, cell) pairs go to their super-cell's range.
template <bool SCATTER>
__global__ __launch_bounds__(256) void coarse_kernel(BinSide A, BinSide B, int blocks_a, int nb,
                                                     const BatchGrid* __restrict__ grids) {
  __shared__ int hist[COARSE_BINS];
  __shared__ int s_info[4];
  const bool second = (int)blockIdx.x >= blocks_a;
  const BinSide& S = second ? B : A;
  const int i0 = ((int)blockIdx.x - (second ? blocks_a : 0)) * COARSE_PTS;
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int last = min(i0 + COARSE_PTS, S.n) - 1;
    const int blo = find_batch(S.off, nb, i0), bhi = find_batch(S.off, nb, last);
    const BatchGrid& gh = grids[bhi];
    s_info[0] = blo;
    s_info[1] = bhi;
    s_info[2] = grids[blo].sup_base;
    s_info[3] = gh.sup_base + ((gh.dim[0] * gh.dim[1] * gh.dim[2] + SUP_CELLS - 1) >> SUP_SHIFT);
  }
  __syncthreads();
  const int blo = s_info[0], bhi = s_info[1], smin = s_info[2], smax = s_info[3];
  const bool in_lds = smax - smin <= COARSE_BINS;  // else: a global atomic per point (clouds with > 2 M cells per block span)
  if (in_lds)
    for (int k = tid; k < smax - smin; k += 256) hist[k] = 0;
  __syncthreads();
  constexpr int PER = COARSE_PTS / 256;
  int sup[PER], rk[PER], cc[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = i0 + k * 256 + tid;
    sup[k] = -1;
    rk[k] = 0;
    cc[k] = 0;
    if (i < S.n) {
      int b = blo;
      if (bhi != blo) b = bhi == blo + 1 ? (i >= S.off[bhi] ? bhi : blo) : find_batch(S.off, nb, i);
      const BatchGrid& g = grids[b];
      int c;
      if (SCATTER) {
        c = S.cell[i];
      } else {
        c = clamped_cell(g, S.pts[3 * (int64_t)i], S.pts[3 * (int64_t)i + 1], S.pts[3 * (int64_t)i + 2]);
        S.cell[i] = c;
      }
      cc[k] = c;
      sup[k] = g.sup_base + ((c - g.cell_base) >> SUP_SHIFT);
      if (in_lds) {
        if (SCATTER) rk[k] = atomicAdd(&hist[sup[k] - smin], 1);
        else atomicAdd(&hist[sup[k] - smin], 1);
      } else {
        if (SCATTER) rk[k] = atomicAdd(&S.sup_cur[sup[k]], 1);
        else atomicAdd(&S.sup_cnt[sup[k]], 1);
      }
    }
  }
  __syncthreads();
  if (in_lds) {
    for (int k = tid; k < smax - smin; k += 256) {
      const int cnt = hist[k];
      if (cnt) {
        if (SCATTER) hist[k] = atomicAdd(&S.sup_cur[smin + k], cnt);  // the block's span inside the super-cell
        else atomicAdd(&S.sup_cnt[smin + k], cnt);
      }
    }
  }
  if (!SCATTER) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = i0 + k * 256 + tid;
    if (sup[k] >= 0) {
      const int dst = S.sup_start[sup[k]] + (in_lds ? hist[sup[k] - smin] : 0) + rk[k];
      S.pairs[dst] = make_int2(i, cc[k]);
    }
  }
}

// exclusive scan of the super-cell counts (one block per side; total_sup is only known on the device)
__global__ __launch_bounds__(1024) void sup_scan_kernel(BinSide A, BinSide B, int first_side, const RadiusHdr* __restrict__ hdr) {
  const BinSide& S = ((int)blockIdx.x + first_side) ? B : A;
  __shared__ int wsum[1024 / WAVE];
  __shared__ int s_carry;
  const int n = hdr->total_sup, tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const int v = i < n ? S.sup_cnt[i] : 0;
    const int inc = wave_incl_scan_add_dpp(v);
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    int base = s_carry, tot = 0;
#pragma unroll
    for (int k = 0; k < 1024 / WAVE; ++k) {
      const int x = wsum[k];
      if (k < w) base += x;
      tot += x;
    }
    if (i < n) S.sup_start[i] = base + inc - v;
    __syncthreads();
    if (tid == 0) s_carry += tot;
    __syncthreads();
  }
  if (tid == 0) S.sup_start[n] = s_carry;
}

// one workgroup per super-cell (grid-stride: the number of super-cells is only known on the device): LDS histogram of its
// <= SUP_CELLS cells, scan -> cell starts, scatter into cell order.  A thread keeps up to FINE_PER of the super-cell's points
// in registers: the (point, cell) pairs are read once and the coordinates are requested before the LDS work starts.
constexpr int FINE_PER = 8;

__global__ __launch_bounds__(256) void fine_kernel(BinSide A, BinSide B, int blocks_a, int nb,
                                                   const BatchGrid* __restrict__ grids, const int32_t* __restrict__ sup_off,
                                                   const RadiusHdr* __restrict__ hdr) {
  __shared__ int hist[SUP_CELLS];
  __shared__ int wsum[256 / WAVE];
  __shared__ float4 stage[256 * FINE_PER];  // the super-cell in cell order: leaves as coalesced copies (records + planes)
  const bool second = (int)blockIdx.x >= blocks_a;
  const BinSide& S = second ? B : A;
  const int stride = second ? (int)gridDim.x - blocks_a : blocks_a;
  const int total_sup = hdr->total_sup;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
  for (int sc = (int)blockIdx.x - (second ? blocks_a : 0); sc < total_sup; sc += stride) {
    const int a = S.sup_start[sc], e = S.sup_start[sc + 1];
    const int b = find_batch(sup_off, nb, sc);
    const BatchGrid& g = grids[b];
    const int ls = sc - g.sup_base;
    const int first = g.cell_base + ls * SUP_CELLS;
    const int ncell = min(SUP_CELLS, g.dim[0] * g.dim[1] * g.dim[2] - ls * SUP_CELLS);
    hist[tid] = 0;
    hist[tid + 256] = 0;
    const bool in_regs = e > a && e - a <= 256 * FINE_PER;
    int2 pr[FINE_PER];
    float cx[FINE_PER], cy[FINE_PER], cz[FINE_PER];
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < FINE_PER; ++k) pr[k] = S.pairs[min(a + k * 256 + tid, e - 1)];
#pragma unroll
      for (int k = 0; k < FINE_PER; ++k) {
        const float* src = S.pts + 3 * (int64_t)pr[k].x;
        cx[k] = src[0];
        cy[k] = src[1];
        cz[k] = src[2];
      }
    }
    __syncthreads();
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < FINE_PER; ++k)
        if (a + k * 256 + tid < e) atomicAdd(&hist[pr[k].y - first], 1);
    } else {
      for (int p = a + tid; p < e; p += 256) atomicAdd(&hist[S.pairs[p].y - first], 1);
    }
    __syncthreads();
    const int v0 = hist[2 * tid], v1 = hist[2 * tid + 1];
    const int inc = wave_incl_scan_add_dpp(v0 + v1);
    if (lane == WAVE - 1) wsum[w] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < 256 / WAVE; ++k)
      if (k < w) base += wsum[k];
    const int ex = base + inc - (v0 + v1);
    hist[2 * tid] = ex;  // becomes the cursor of the cell
    hist[2 * tid + 1] = ex + v0;
    if (S.cell_start) {
      if (2 * tid < ncell) S.cell_start[first + 2 * tid] = a + ex;
      if (2 * tid + 1 < ncell) S.cell_start[first + 2 * tid + 1] = a + ex + v0;
      if (sc == total_sup - 1 && tid == 0) S.cell_start[first + ncell] = e;  // end of the last cell of the last cloud
    }
    __syncthreads();
    // order inside a cell: arrival (the search results do not depend on it)
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < FINE_PER; ++k)
        if (a + k * 256 + tid < e) {
          const int slot = atomicAdd(&hist[pr[k].y - first], 1);
          stage[slot] = make_float4(cx[k], cy[k], cz[k], __int_as_float(pr[k].x));
        }
      __syncthreads();
      // (scattering 16 + 3 x 4 bytes per point straight to global memory took 49 us of the 8 x 200 k binning; staged: 29)
      for (int p = tid; p < e - a; p += 256) {
        const float4 v = stage[p];
        S.sorted[a + p] = v;
        if (S.plane_x) {
          S.plane_x[a + p] = v.x;
          S.plane_y[a + p] = v.y;
          S.plane_z[a + p] = v.z;
        }
      }
    } else {
      for (int p = a + tid; p < e; p += 256) {
        const int2 q = S.pairs[p];
        const int slot = atomicAdd(&hist[q.y - first], 1);
        const float* src = S.pts + 3 * (int64_t)q.x;
        S.sorted[a + slot] = make_float4(src[0], src[1], src[2], __int_as_float(q.x));
        if (S.plane_x) {
          S.plane_x[a + slot] = src[0];
          S.plane_y[a + slot] = src[1];
          S.plane_z[a + slot] = src[2];
        }
      }
    }
    __syncthreads();  // hist is cleared by the next super-cell of this workgroup
  }
}

// ---------------------------------------------------------------- candidate traversal
// A block owns RQ consecutive cell-ordered queries and runs 3*RQ threads: thread (j, slot) walks
// the three (dy, dz = j-1) bands of query `slot`, so a wave holds 64 neighbouring queries looking
// at the same z-slab.  Cells are numbered x-fastest, hence the union of the block's 27-cell
// neighbourhoods is nine (dy,dz) "bands", each a CONTIGUOUS range of the cell-sorted support array.
// The block stages those ranges in LDS with coalesced float4 loads (falls back to direct global
// reads if they do not fit) and every thread then walks its own candidates out of LDS, four
// independent ds_read_b128 in flight per step.
//   COUNT pass: hits per (query, z-slab) -> q_cnt[3][nq]; per-block max / sum -> blk_stats
//   FILL  pass: phase A appends (dist,index) keys unsorted into per-query LDS segments (the slab
//               sub-counts give every thread a private sub-segment: no atomics);
//               phase B gives each hit one thread, ranks it inside its segment (branch-free
//               counting; segment reads are LDS broadcasts) and stores it straight to its final
//               slot out[query][rank]; padding is written one row per wave.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NBAND = 9;
constexpr int NSUB = 3;  // threads per query (one per z-slab)
constexpr int GB = 8;    // hit loads in flight per thread in the FILL gather

template <int RQ>
struct TravLds {
  static constexpr int THREADS = NSUB * RQ;
  static constexpr int STAGE_CAP = 12 * RQ;  // candidates the block can stage
  // ints: offs[RQ+1], orig[RQ], wsum[THREADS/64], sub[3*RQ], band_lo[9], band_hi[9], band_base[10]
  static constexpr int TABLE_MAX = 256;  // clouds whose offsets / grids are cached in LDS (sized per launch)
  static constexpr int N_INTS = (RQ + 1) + RQ + NSUB * RQ + 9 + 9 + 10 + THREADS / WAVE;
  static constexpr size_t TABLE_OFF = (size_t)(N_INTS * 4 + 15) / 16 * 16;
  // the FILL pass only needs offs, orig and wsum (laid out first): its hit segments start right after them
  static constexpr size_t FILL_OFF = (size_t)(((RQ + 1) + RQ + THREADS / WAVE) * 4 + 15) / 16 * 16;
  // COUNT pass: [int tables | q offsets of `tcap` clouds | their grids | candidate planes]
  static __host__ __device__ size_t tables_bytes(int tcap) { return tcap > 0 ? ((size_t)(tcap + 1) * 4 + 15) / 16 * 16 + (size_t)tcap * sizeof(BatchGrid) : 0; }
  static constexpr size_t STAGE_BYTES = (size_t)STAGE_CAP * 12;  // three coordinate planes
  static size_t count_bytes(int tcap) { return TABLE_OFF + tables_bytes(tcap) + STAGE_BYTES; }
  // FILL: slots = hits + at most one pad slot per query, rounded to 16 so every block's key array stays 16-B aligned
  static int64_t slots(int64_t max_block_hits) { return (max_block_hits + RQ + 15) / 16 * 16; }
  static size_t total(int64_t slots) { return FILL_OFF + (size_t)slots * 9; }  // int tables + keys (8 B) + row ids (1 B)
};

template <int RQ, bool FILL, bool HITS_IN_LDS>
__global__ __launch_bounds__(NSUB* RQ) __attribute__((amdgpu_waves_per_eu(8, 8))) void traverse_kernel(
    const float4* __restrict__ sorted_q, int nq, const int32_t* __restrict__ q_off, int nb,
    const BatchGrid* __restrict__ grids, const int32_t* __restrict__ start_s,
    const float4* __restrict__ sorted_s, float r2, int32_t* __restrict__ q_cnt, int2* __restrict__ q_rng,
    unsigned long long* __restrict__ q_mask, int32_t* __restrict__ blk_stats, int width, int row_stride, int64_t pad_value,
    int64_t* __restrict__ out, int max_block_hits, unsigned long long* __restrict__ g_hits, unsigned char* __restrict__ g_rows, int mono) {
  using L = TravLds<RQ>;
  static_assert(RQ % WAVE == 0 && RQ <= 256, "row ids are bytes; waves must not straddle slabs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* offs = reinterpret_cast<int*>(smem);
  int* orig = offs + (RQ + 1);
  int* wsum = orig + RQ;
  int* sub = wsum + L::THREADS / WAVE;  // [NSUB][RQ]
  int* band_lo = sub + NSUB * RQ;
  int* band_hi = band_lo + NBAND;
  int* band_base = band_hi + NBAND;
  // COUNT pass only: per-cloud tables cached in LDS so the per-query setup is not a chain of
  // dependent global round trips (query -> cloud id -> grid -> cell starts)
  const int tcap = FILL ? 0 : (nb <= L::TABLE_MAX ? nb : 0);
  int* s_qoff = reinterpret_cast<int*>(smem + L::TABLE_OFF);
  BatchGrid* s_grids = reinterpret_cast<BatchGrid*>(smem + L::TABLE_OFF + ((size_t)(tcap + 1) * 4 + 15) / 16 * 16);
  float4* stage = reinterpret_cast<float4*>(smem + L::TABLE_OFF + L::tables_bytes(tcap));
  // FILL keeps no candidate stage: its hit segments start right after the int tables
  unsigned long long* hits = HITS_IN_LDS ? reinterpret_cast<unsigned long long*>(smem + L::FILL_OFF)
                                         : g_hits + (int64_t)blockIdx.x * max_block_hits;
  unsigned char* rows = HITS_IN_LDS
                            ? reinterpret_cast<unsigned char*>(smem + L::FILL_OFF + (size_t)max_block_hits * 8)
                            : g_rows + (int64_t)blockIdx.x * max_block_hits;

  const int tid = threadIdx.x;
  const int slot = tid % RQ, j = tid / RQ;  // query slot in block, z-slab
  // XCD-aware block order: the dispatcher places block b on XCD b % 8 (speed only, never
  // correctness).  Give each XCD one CONTIGUOUS eighth of the cell-ordered queries so the candidate
  // bands of neighbouring blocks (which overlap ~9x) are served by that XCD's own 4 MiB L2 instead
  // of being re-fetched from Infinity Cache by all eight.
  const int nblk = (nq + RQ - 1) / RQ;
  const int per_xcd = gridDim.x / 8;  // the grid is padded to a multiple of 8 blocks
  const int blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (blk >= nblk) return;
  const int t = blk * RQ + slot;
  const int lane = tid & (WAVE - 1);
  const bool valid = t < nq;

  if (tid < NBAND) {
    band_lo[tid] = 0x7fffffff;
    band_hi[tid] = 0;
  }
  const bool tables_in_lds = tcap > 0;
  if (tables_in_lds) {
    for (int i = tid; i <= nb; i += L::THREADS) s_qoff[i] = q_off[i];
    const int4* gsrc = reinterpret_cast<const int4*>(grids);
    int4* gdst = reinterpret_cast<int4*>(s_grids);
    for (int i = tid; i < nb * 4; i += L::THREADS) gdst[i] = gsrc[i];
  }
  int my_off = 0;
  float4 qp = make_float4(0.f, 0.f, 0.f, 0.f);
  int p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0};
  unsigned long long fill_bits = 0ull;
  if (FILL) {
    // every slab group redundantly scans the per-query totals (two waves each; no cross-group sync)
    int c[NSUB] = {0, 0, 0};
    if (valid) {
#pragma unroll
      for (int i = 0; i < NSUB; ++i) c[i] = q_cnt[(int64_t)i * nq + t];
      // everything else this thread needs from the COUNT pass is requested NOW, so the block pays one global
      // round trip for (counts, query, ranges, hit mask) instead of two separated by the barrier below
      qp = sorted_q[t];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int2 r = q_rng[(int64_t)(i * NSUB + j) * nq + t];
        p0[i] = r.x;
        p1[i] = r.y;
      }
      fill_bits = q_mask[(int64_t)j * nq + t];
    }
    const int tot = c[0] + c[1] + c[2];
    const int tot2 = (tot + 1) & ~1;  // segments start on even slots: the rank loop reads two keys per ds_read_b128
    const int inc = wave_incl_scan_add_dpp(tot2);
    if (lane == WAVE - 1) wsum[tid / WAVE] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < RQ / WAVE; ++i)
      if (i < slot / WAVE) base += wsum[j * (RQ / WAVE) + i];
    const int q_start = base + inc - tot2;
    my_off = q_start + (j > 0 ? c[0] : 0) + (j > 1 ? c[1] : 0);
    if (j == 0) {
      offs[slot] = q_start;
      if (slot == RQ - 1) offs[RQ] = q_start + tot2;
      if (tot2 != tot) {  // pad slot: larger than every real key, skipped by the rank phase
        hits[q_start + tot] = ~0ull;
        rows[q_start + tot] = 0xff;
      }
    }
  } else {
    __syncthreads();
  }

  // ---- per-thread candidate ranges (global positions in sorted_s) for bands (dy, dz = j-1):
  //      computed by the COUNT pass and stored; the FILL pass just reloads them (one coalesced trip)
  if (valid) {
    if (FILL) {
      if (j == 0) orig[slot] = __float_as_int(qp.w);
    } else {
      qp = sorted_q[t];
      int b;
      BatchGrid g;
      if (tables_in_lds) {
        b = find_batch(s_qoff, nb, __float_as_int(qp.w));
        g = s_grids[b];
      } else {
        b = find_batch(q_off, nb, __float_as_int(qp.w));
        g = grids[b];
      }
      const double ux = cell_coord(qp.x, g.org[0], g.inv_cell_x), kx = (double)g.xk;
      const double uy = cell_coord(qp.y, g.org[1], g.inv_cell);
      const double cz = cell_coord(qp.z, g.org[2], g.inv_cell) + (double)(j - 1);
      const double tx = (double)(g.dim[0] - 1), ty = (double)(g.dim[1] - 1), tz = (double)(g.dim[2] - 1);
      // the comparisons are written so that NaN coordinates give "no candidates"
      if ((ux + kx >= 0.0) && (ux - kx <= tx) && cz >= 0.0 && cz <= tz) {
        const int lx = (int)fmin(fmax(ux - kx, 0.0), tx);
        const int hx = (int)fmin(fmax(ux + kx, 0.0), tx);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double cy = uy + (double)(i - 1);
          if (cy >= 0.0 && cy <= ty) {
            const int base = g.cell_base + g.dim[0] * ((int)cy + g.dim[1] * (int)cz);
            p0[i] = start_s[base + lx];
            p1[i] = start_s[base + hx + 1];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) q_rng[(int64_t)(i * NSUB + j) * nq + t] = make_int2(p0[i], p1[i]);
    }
  } else if (FILL && j == 0) {
    orig[slot] = -1;
  }
  int n = 0;
  if (!FILL) {
    // ---- block-wide extent of every band (waves are slab-uniform: band index = 3*j + i)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool has = p1[i] > p0[i];
      int lo, hi;
      if (mono) {
        // self-search: queries are in cell order and every range comes from the query's own cell, so p0 and p1 are
        // non-decreasing along the wave -- the extent is (first valid lane's p0, last valid lane's p1)
        const unsigned long long m = __ballot(has);
        lo = 0x7fffffff;
        hi = 0;
        if (m) {
          lo = __builtin_amdgcn_readlane(p0[i], __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1));
          hi = __builtin_amdgcn_readlane(p1[i], __builtin_amdgcn_readfirstlane(63 - __clzll((long long)m)));
        }
      } else {
        lo = wave_min_i32_dpp(has ? p0[i] : 0x7fffffff);
        hi = wave_max_i32_dpp(has ? p1[i] : 0);
      }
      if (lane == 0 && hi > 0) {
        atomicMin(&band_lo[3 * j + i], lo);
        atomicMax(&band_hi[3 * j + i], hi);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int k = 0; k < NBAND; ++k) {
        band_base[k] = acc;
        acc += band_hi[k] > band_lo[k] ? band_hi[k] - band_lo[k] : 0;
      }
      band_base[NBAND] = acc;
    }
    __syncthreads();
    const bool staged = band_base[NBAND] <= L::STAGE_CAP;
    // candidates are staged as three coordinate planes (the index is not needed to COUNT), so a thread can
    // pull two neighbours per plane into one 64-bit register pair and test them with packed fp32 math
    float* sx = reinterpret_cast<float*>(stage);
    float* sy = sx + L::STAGE_CAP;
    float* sz = sy + L::STAGE_CAP;
    if (staged) {
      // one flat pass over the union of the nine bands: every thread issues ALL its loads (<= 4) before the
      // first LDS write, so the block pays one global round trip here instead of one per band
      const int total = band_base[NBAND];
      int bl[NBAND], bs[NBAND];
#pragma unroll
      for (int k = 0; k < NBAND; ++k) {
        bl[k] = band_lo[k];
        bs[k] = band_base[k];
      }
      constexpr int PER = L::STAGE_CAP / L::THREADS;
      float4 v[PER];
      // unconditional loads on a clamped index (pad_value = number of supports): behind `if (f < total)` the compiler
      // keeps every load behind the previous one's use -- four memory round trips instead of one
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = tid + u * L::THREADS;
        unsigned src = (unsigned)bl[0] + (unsigned)f;
#pragma unroll
        for (int k = 1; k < NBAND; ++k) src = f >= bs[k] ? (unsigned)bl[k] + (unsigned)(f - bs[k]) : src;
        src = f < total ? src : 0u;
        v[u] = sorted_s[min(src, (unsigned)((int)pad_value - 1))];
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = tid + u * L::THREADS;
        if (f < total) {
          sx[f] = v[u].x;
          sy[f] = v[u].y;
          sz[f] = v[u].z;
        }
      }
      __syncthreads();
    }
    // ---- walk every candidate; remember the hits as a bit mask (bit = position in this thread's
    //      enumeration order) so the FILL pass only ever touches the ~16 % that matter
    unsigned long long mask = 0ull;
    int bitpos = 0;
    if (valid && staged) {
      const f32x2 qx = {qp.x, qp.x}, qy = {qp.y, qp.y}, qz = {qp.z, qp.z};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int rel = band_base[3 * j + i] - band_lo[3 * j + i];
        int p = p0[i] + rel;
        const int e = p1[i] + rel;
        for (; p + 4 <= e; p += 4) {
          const f32x2 xa = {sx[p], sx[p + 1]}, xb = {sx[p + 2], sx[p + 3]};
          const f32x2 ya = {sy[p], sy[p + 1]}, yb = {sy[p + 2], sy[p + 3]};
          const f32x2 za = {sz[p], sz[p + 1]}, zb = {sz[p + 2], sz[p + 3]};
          // nanoflann.hpp:432-440: result += diff*diff for x, y, z starting from 0 (two lanes per op)
          const f32x2 dxa = qx - xa, dya = qy - ya, dza = qz - za;
          const f32x2 dxb = qx - xb, dyb = qy - yb, dzb = qz - zb;
          const f32x2 da = (dxa * dxa + dya * dya) + dza * dza;
          const f32x2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;
          const unsigned hb = (da.x < r2 ? 1u : 0u) | (da.y < r2 ? 2u : 0u) | (db.x < r2 ? 4u : 0u) | (db.y < r2 ? 8u : 0u);
          if (bitpos < 64) mask |= (unsigned long long)hb << bitpos;
          n += __popc(hb);
          bitpos += 4;
        }
        for (; p < e; ++p) {
          const float dx = qp.x - sx[p], dy = qp.y - sy[p], dz = qp.z - sz[p];
          const float d = (dx * dx + dy * dy) + dz * dz;
          const bool hit = d < r2;
          if (hit && bitpos < 64) mask |= 1ull << bitpos;
          n += hit ? 1 : 0;
          ++bitpos;
        }
      }
    } else if (valid) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        for (int p = p0[i]; p < p1[i]; ++p) {
          const float4 sp = sorted_s[p];
          const float dx = qp.x - sp.x, dy = qp.y - sp.y, dz = qp.z - sp.z;
          const float d = (dx * dx + dy * dy) + dz * dz;
          const bool hit = d < r2;
          if (hit && bitpos < 64) mask |= 1ull << bitpos;
          n += hit ? 1 : 0;
          ++bitpos;
        }
    }
    if (valid) q_mask[(int64_t)j * nq + t] = mask;
  } else if (valid) {
    // ---- FILL: gather only the hits (bit mask from the COUNT pass), eight loads in flight;
    //      threads with more than 64 candidates re-walk everything
    const int len0 = p1[0] - p0[0], len1 = p1[1] - p0[1], len2 = p1[2] - p0[2];
    auto emit = [&](const float4 sp) {
      const float dx = qp.x - sp.x;
      const float dy = qp.y - sp.y;
      const float dz = qp.z - sp.z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d < r2) {
        // key orders by (distance, index); d >= 0 so its bit pattern is monotone
        hits[my_off + n] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)__float_as_int(sp.w);
        rows[my_off + n] = (unsigned char)slot;
        ++n;
      }
    };
    if (len0 + len1 + len2 <= 64) {
      unsigned long long bits = fill_bits;
      while (bits) {
        int pos[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          pos[u] = -1;
          if (bits) {
            const int bpos = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            // enumeration order: band 0, then band 1, then band 2
            pos[u] = bpos < len0 ? p0[0] + bpos : (bpos < len0 + len1 ? p0[1] + (bpos - len0) : p0[2] + (bpos - len0 - len1));
          }
        }
        // unconditional loads (a spent slot re-reads support 0): behind a branch the compiler waits for every load before
        // it issues the next one, and the point of this loop is GB random reads in flight
        float4 sp[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) sp[u] = sorted_s[max(pos[u], 0)];
#pragma unroll
        for (int u = 0; u < GB; ++u)
          if (pos[u] >= 0) emit(sp[u]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        for (int p = p0[i]; p < p1[i]; ++p) emit(sorted_s[p]);
    }
  }

  if (!FILL) {
    if (valid) q_cnt[(int64_t)j * nq + t] = n;
    sub[tid] = n;
    __syncthreads();
    if (tid < RQ) {
      const int tot = sub[tid] + sub[RQ + tid] + sub[2 * RQ + tid];
      const int mx = wave_max_i32_dpp(tot), sm = wave_sum_i32_dpp(tot);
      if (lane == 0) {
        wsum[tid / WAVE] = mx;
        wsum[RQ / WAVE + tid / WAVE] = sm;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int mx = 0, sm = 0;
#pragma unroll
      for (int i = 0; i < RQ / WAVE; ++i) {
        mx = max(mx, wsum[i]);
        sm += wsum[RQ / WAVE + i];
      }
      blk_stats[2 * blk] = mx;      // reduced by reduce_stats_kernel: no same-address
      blk_stats[2 * blk + 1] = sm;  // global atomics (they cost ~11 ns EACH when contended)
    }
    return;
  }

  __syncthreads();
  // ---- phase B: one thread per hit, rank inside its segment, store to the final slot
  const int total_hits = offs[RQ];
  for (int e = tid; e < total_hits; e += L::THREADS) {
    const int r = rows[e];
    if (r == 0xff) continue;  // pad slot
    const int a = offs[r], len = offs[r + 1] - a;  // both even
    const unsigned long long key = hits[e];
    const ulonglong2* seg = reinterpret_cast<const ulonglong2*>(hits + a);
    int rank = 0;
    int jj = 0;
    for (; jj + 4 <= len / 2; jj += 4) {  // eight keys per step, four independent ds_read_b128 in flight
      ulonglong2 hk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) hk[u] = seg[jj + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) rank += (hk[u].x < key ? 1 : 0) + (hk[u].y < key ? 1 : 0);
    }
    for (; jj < len / 2; ++jj) {
      const ulonglong2 h = seg[jj];
      rank += (h.x < key ? 1 : 0) + (h.y < key ? 1 : 0);
    }
    if (rank < width) out[(int64_t)orig[r] * row_stride + rank] = (int64_t)(unsigned int)(key & 0xffffffffull);
  }
  // ---- padding: one row per wave iteration, lanes along the row
  const int rows_here = min(RQ, nq - blk * RQ);
  for (int r = tid / 32; r < rows_here; r += L::THREADS / 32) {  // half a wave per row
    int cnt = offs[r + 1] - offs[r];
    if (cnt > 0 && rows[offs[r + 1] - 1] == 0xff) --cnt;  // the segment ends in a pad slot
    int64_t* row = out + (int64_t)orig[r] * row_stride;
    for (int c = min(cnt, width) + (lane & 31); c < row_stride; c += 32) row[c] = pad_value;
  }
}

// ---------------------------------------------------------------- single pass for a width known before the launch
// radius_search(..., neighbor_limit) (modules/ops/radius_search.py:7-27) keeps min(max_count, neighbor_limit) columns, so the
// caller can allocate (nq, limit) rows BEFORE anything is counted and one kernel does the whole search:
//   set-up, staging  as in the COUNT pass above
//   tests            hits are remembered in two 32-bit masks per thread (even / odd candidates of its enumeration)
//   scan             hit counts -> per-query segments in the block's key area (LDS)
//   decode           every thread walks its masks, one even and one odd hit per step, and leaves (query slot, staged
//                    position) words in its part of the segment -- no arithmetic in the loop whose trip count diverges
//   keys             one thread per hit (balanced): distance bits and support index from the staged planes
//   ranking          one thread per hit: rank = number of smaller distance words in the segment (32-bit compares, four keys
//                    per ds_read_b128).  The index goes to row[rank] in an LDS row buffer with ds_min: equal distances
//                    collide there, leave a hole behind them, and only such rows are ranked again on (distance, index)
//   rows             whole rows leave as contiguous 16-byte pieces
// Nothing per query goes through global memory in between (the two-pass path writes and re-reads 180 bytes of ranges /
// masks / counts per query) and the host does not sit between two launches.
//   blk_stats[2 blk]     = largest hit count of a query in the block   (max -> the width the reference would return)
//   blk_stats[2 blk + 1] = 1 if a single query had more hits than the block's key area holds (the caller then repeats
//                          the search on the two-pass path)
// A block whose hits do not fit its key area at once works through its queries in groups (direct stores, exact compare).
template <int RQ>
struct FusedLds {
  static constexpr int THREADS = NSUB * RQ;
  static constexpr int STAGE_CAP = 12 * RQ;
  static constexpr int TABLE_MAX = 256;
  // ints: offs[RQ+1], orig[RQ], qtot[RQ], wsum[2 * THREADS/64], sub[3*RQ], band_lo[9], band_hi[9], band_base[10], misc[4],
  //       tie flags[RQ]
  static constexpr int N_INTS = (RQ + 1) + RQ + RQ + 2 * (THREADS / WAVE) + NSUB * RQ + 9 + 9 + 10 + 4 + RQ;
  static constexpr size_t QBUF_OFF = (size_t)(N_INTS * 4 + 15) / 16 * 16;  // float4 per query slot
  static constexpr size_t STAGE_OFF = QBUF_OFF + (size_t)RQ * 16;
  static size_t region_bytes(int width) {  // candidate planes x, y, z, index (+ slack for the 4-wide tail reads); the row
    const size_t st = (size_t)STAGE_CAP * 16 + 16, rb = ((size_t)RQ * width * 4 + 15) / 16 * 16;  // buffer takes their place
    return st > rb ? st : rb;
  }
  static size_t tables_bytes(int tcap) { return tcap > 0 ? ((size_t)(tcap + 1) * 4 + 15) / 16 * 16 + (size_t)tcap * sizeof(BatchGrid) : 0; }
  static size_t hits_bytes(int cap) { return (size_t)(cap + 16) * 9; }  // distance words, (slot, position) / index words, row bytes
  static size_t total(int width, int cap, int tcap) {
    const size_t hits = hits_bytes(cap), tb = tables_bytes(tcap);
    return STAGE_OFF + region_bytes(width) + (hits > tb ? hits : tb);
  }
};

template <int RQ>
__global__ __launch_bounds__(NSUB* RQ) void fused_kernel(
    const float4* __restrict__ sorted_q, int nq, const int32_t* __restrict__ q_off, int nb,
    const BatchGrid* __restrict__ grids, const int32_t* __restrict__ start_s, const float4* __restrict__ sorted_s, int ns_total,
    float r2, int32_t* __restrict__ blk_stats, int width, int64_t pad_value, int64_t* __restrict__ out, int cap,
    int region_bytes, int mono) {
  using L = FusedLds<RQ>;
  static_assert(RQ % WAVE == 0 && RQ <= 256, "row ids are bytes; waves must not straddle slabs");
  constexpr unsigned PADMARK = 0xffffffffu;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* offs = reinterpret_cast<int*>(smem);
  int* orig = offs + (RQ + 1);
  int* qtot = orig + RQ;
  int* wsum = qtot + RQ;
  int* sub = wsum + 2 * (L::THREADS / WAVE);  // [NSUB][RQ]
  int* band_lo = sub + NSUB * RQ;
  int* band_hi = band_lo + NBAND;
  int* band_base = band_hi + NBAND;
  int* misc = band_base + NBAND + 1;  // [0] group search, [1] number of rows with equal distances
  int* tie_rows = misc + 4;
  float4* qbuf = reinterpret_cast<float4*>(smem + L::QBUF_OFF);
  float* sx = reinterpret_cast<float*>(smem + L::STAGE_OFF);
  float* sy = sx + L::STAGE_CAP;
  float* sz = sy + L::STAGE_CAP;
  int* si = reinterpret_cast<int*>(sz + L::STAGE_CAP);
  unsigned int* rowbuf = reinterpret_cast<unsigned int*>(smem + L::STAGE_OFF);  // takes the planes' place after the keys pass
  char* hreg = smem + L::STAGE_OFF + region_bytes;
  unsigned int* hd = reinterpret_cast<unsigned int*>(hreg);      // distance bits per hit slot
  unsigned int* hm = hd + (cap + 16);                            // (slot << 16 | staged position), then the support index
  unsigned char* hrow = reinterpret_cast<unsigned char*>(hm + (cap + 16));
  const int dummy = cap + 8;  // a slot nobody reads: the target of the decode's "no hit" lanes
  // per-cloud tables for the set-up live where the keys go later
  const int tcap = nb <= L::TABLE_MAX ? nb : 0;
  int* s_qoff = reinterpret_cast<int*>(hreg);
  BatchGrid* s_grids = reinterpret_cast<BatchGrid*>(hreg + ((size_t)(tcap + 1) * 4 + 15) / 16 * 16);

  const int tid = threadIdx.x;
  const int slot = tid % RQ, j = tid / RQ;
  const int nblk = (nq + RQ - 1) / RQ;
  const int per_xcd = gridDim.x / 8;
  const int blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // one contiguous eighth of the cell-ordered queries per XCD
  if (blk >= nblk) return;
  const int t = blk * RQ + slot;
  const int lane = tid & (WAVE - 1);
  const bool valid = t < nq;

  if (tid < NBAND) {
    band_lo[tid] = 0x7fffffff;
    band_hi[tid] = 0;
  }
  if (tid == 0) misc[1] = 0;
  if (tid < RQ) tie_rows[tid] = 0;
  const bool tables_in_lds = tcap > 0;
  if (tables_in_lds) {
    for (int i = tid; i <= nb; i += L::THREADS) s_qoff[i] = q_off[i];
    const int4* gsrc = reinterpret_cast<const int4*>(grids);
    int4* gdst = reinterpret_cast<int4*>(s_grids);
    for (int i = tid; i < nb * 4; i += L::THREADS) gdst[i] = gsrc[i];
  }
  float4 qp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) qp = sorted_q[t];
  __syncthreads();
  int p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0};
  if (valid) {
    int b;
    BatchGrid g;
    if (tables_in_lds) {
      b = find_batch(s_qoff, nb, __float_as_int(qp.w));
      g = s_grids[b];
    } else {
      b = find_batch(q_off, nb, __float_as_int(qp.w));
      g = grids[b];
    }
    const double ux = cell_coord(qp.x, g.org[0], g.inv_cell_x), kx = (double)g.xk;
    const double uy = cell_coord(qp.y, g.org[1], g.inv_cell);
    const double cz = cell_coord(qp.z, g.org[2], g.inv_cell) + (double)(j - 1);
    const double tx = (double)(g.dim[0] - 1), ty = (double)(g.dim[1] - 1), tz = (double)(g.dim[2] - 1);
    if ((ux + kx >= 0.0) && (ux - kx <= tx) && cz >= 0.0 && cz <= tz) {  // NaN coordinates: no candidates
      const int lx = (int)fmin(fmax(ux - kx, 0.0), tx);
      const int hx = (int)fmin(fmax(ux + kx, 0.0), tx);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double cy = uy + (double)(i - 1);
        if (cy >= 0.0 && cy <= ty) {
          const int base = g.cell_base + g.dim[0] * ((int)cy + g.dim[1] * (int)cz);
          p0[i] = start_s[base + lx];
          p1[i] = start_s[base + hx + 1];
        }
      }
    }
    if (j == 0) {
      orig[slot] = __float_as_int(qp.w);
      qbuf[slot] = qp;
    }
  } else if (j == 0) {
    orig[slot] = -1;
  }
  // ---- block-wide extent of every band (waves are slab-uniform: band index = 3*j + i)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool has = p1[i] > p0[i];
    int lo, hi;
    if (mono) {
      const unsigned long long m = __ballot(has);
      lo = 0x7fffffff;
      hi = 0;
      if (m) {
        lo = __builtin_amdgcn_readlane(p0[i], __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1));
        hi = __builtin_amdgcn_readlane(p1[i], __builtin_amdgcn_readfirstlane(63 - __clzll((long long)m)));
      }
    } else {
      lo = wave_min_i32_dpp(has ? p0[i] : 0x7fffffff);
      hi = wave_max_i32_dpp(has ? p1[i] : 0);
    }
    if (lane == 0 && hi > 0) {
      atomicMin(&band_lo[3 * j + i], lo);
      atomicMax(&band_hi[3 * j + i], hi);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < NBAND; ++k) {
      band_base[k] = acc;
      acc += band_hi[k] > band_lo[k] ? band_hi[k] - band_lo[k] : 0;
    }
    band_base[NBAND] = acc;
  }
  __syncthreads();
  const bool staged = band_base[NBAND] <= L::STAGE_CAP;
  if (staged) {
    // wave w copies bands w, w + NWV, ...; lanes run over the band's elements.  (A flat pass over the union of the bands had
    // every element find its band with eight compare / select pairs: ~130 instructions per thread for four elements.)  All
    // loads of a wave are issued before its first LDS write, on clamped indices (no load sits behind a branch).
    constexpr int NWV = L::THREADS / WAVE, KMAX = (NBAND + NWV - 1) / NWV, UNR = 2;
    const int wvi = tid / WAVE;
    float4 v[KMAX][UNR];
    int blo[KMAX], blen[KMAX], bdst[KMAX];
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
      const int k = wvi + kk * NWV;
      blo[kk] = 0;
      blen[kk] = 0;
      bdst[kk] = 0;
      if (k < NBAND) {
        const int l0 = band_lo[k], h0 = band_hi[k];
        blen[kk] = h0 > l0 ? h0 - l0 : 0;
        blo[kk] = blen[kk] > 0 ? l0 : 0;
        bdst[kk] = band_base[k];
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        v[kk][u] = sorted_s[min(blo[kk] + u * WAVE + lane, ns_total - 1)];
    }
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int f = u * WAVE + lane;
        if (f < blen[kk]) {
          sx[bdst[kk] + f] = v[kk][u].x;
          sy[bdst[kk] + f] = v[kk][u].y;
          sz[bdst[kk] + f] = v[kk][u].z;
          si[bdst[kk] + f] = __float_as_int(v[kk][u].w);
        }
      }
      for (int f = UNR * WAVE + lane; f < blen[kk]; f += WAVE) {  // a band longer than 128 elements
        const float4 t4 = sorted_s[blo[kk] + f];
        sx[bdst[kk] + f] = t4.x;
        sy[bdst[kk] + f] = t4.y;
        sz[bdst[kk] + f] = t4.z;
        si[bdst[kk] + f] = __float_as_int(t4.w);
      }
    }
    __syncthreads();
  }
  // ---- test every candidate, four per step.  Enumeration slot c = 4 * step + k (k = 0..3; a band's last step is padded);
  //      even slots are remembered in `lo`, odd slots in `hi`: two 32-bit SHIFT REGISTERS -- a hit is the sign bit of
  //      (distance bits - r2 bits) (both are non-negative floats: their bit patterns order like the values, NaN sorts above
  //      everything), shifted in with one v_alignbit; the decode below takes one hit from each side per step
  unsigned lo = 0u, hi = 0u;
  int n = 0;
  int rel[3] = {0, 0, 0};
  if (staged) {
#pragma unroll
    for (int i = 0; i < 3; ++i) rel[i] = band_base[3 * j + i] - band_lo[3 * j + i];
  }
  const int len0 = p1[0] - p0[0], len1 = p1[1] - p0[1], len2 = p1[2] - p0[2];
  const int nit0 = (len0 + 3) >> 2, nit1 = (len1 + 3) >> 2, nit2 = (len2 + 3) >> 2;
  const bool by_mask = staged && (nit0 + nit1 + nit2 <= 16);  // else: counted here, re-walked in the decode
  const unsigned r2b = r2 == r2 ? __float_as_uint(r2) : 0u;      // NaN radius: nothing is a neighbour
  if (valid && by_mask) {
    const f32x2 qx = {qp.x, qp.x}, qy = {qp.y, qp.y}, qz = {qp.z, qp.z};
    auto step4 = [&](int p) {
      const f32x2 xa = {sx[p], sx[p + 1]}, xb = {sx[p + 2], sx[p + 3]};
      const f32x2 ya = {sy[p], sy[p + 1]}, yb = {sy[p + 2], sy[p + 3]};
      const f32x2 za = {sz[p], sz[p + 1]}, zb = {sz[p + 2], sz[p + 3]};
      // nanoflann.hpp:432-440: result += diff*diff for x, y, z starting from 0 (two candidates per op)
      const f32x2 dxa = qx - xa, dya = qy - ya, dza = qz - za;
      const f32x2 dxb = qx - xb, dyb = qy - yb, dzb = qz - zb;
      const f32x2 da = (dxa * dxa + dya * dya) + dza * dza;
      const f32x2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;
      const unsigned t0 = __float_as_uint(da.x) - r2b, t1 = __float_as_uint(da.y) - r2b;
      const unsigned t2 = __float_as_uint(db.x) - r2b, t3 = __float_as_uint(db.y) - r2b;
      lo = __builtin_amdgcn_alignbit(lo, t0, 31);  // (lo << 1) | sign(t0)
      lo = __builtin_amdgcn_alignbit(lo, t2, 31);
      hi = __builtin_amdgcn_alignbit(hi, t1, 31);
      hi = __builtin_amdgcn_alignbit(hi, t3, 31);
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int p = p0[i] + rel[i];
      const int e = p1[i] + rel[i];
      for (; p + 4 <= e; p += 4) step4(p);
      if (p < e) {  // padded last step: 1..3 candidates left (the reads past the band stay inside the planes)
        const int left = e - p;
        // (same arithmetic; slots past the band shift in a zero)
        {
          const f32x2 xa = {sx[p], sx[p + 1]}, xb = {sx[p + 2], sx[p + 3]};
          const f32x2 ya = {sy[p], sy[p + 1]}, yb = {sy[p + 2], sy[p + 3]};
          const f32x2 za = {sz[p], sz[p + 1]}, zb = {sz[p + 2], sz[p + 3]};
          const f32x2 dxa = qx - xa, dya = qy - ya, dza = qz - za;
          const f32x2 dxb = qx - xb, dyb = qy - yb, dzb = qz - zb;
          const f32x2 da = (dxa * dxa + dya * dya) + dza * dza;
          const f32x2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;
          const unsigned t0 = __float_as_uint(da.x) - r2b;
          const unsigned t1 = left >= 2 ? __float_as_uint(da.y) - r2b : 0u;
          const unsigned t2 = left >= 3 ? __float_as_uint(db.x) - r2b : 0u;
          lo = __builtin_amdgcn_alignbit(lo, t0, 31);
          lo = __builtin_amdgcn_alignbit(lo, t2, 31);
          hi = __builtin_amdgcn_alignbit(hi, t1, 31);
          hi = __builtin_amdgcn_alignbit(hi, 0u, 31);
        }
      }
    }
    n = __popc(lo) + __popc(hi);
  } else if (valid && staged) {  // more than 64 enumeration slots: count now, walk again in the decode
#pragma unroll
    for (int i = 0; i < 3; ++i)
      for (int p = p0[i] + rel[i]; p < p1[i] + rel[i]; ++p) {
        const float dx = qp.x - sx[p], dy = qp.y - sy[p], dz = qp.z - sz[p];
        const float d = (dx * dx + dy * dy) + dz * dz;
        n += d < r2 ? 1 : 0;
      }
  } else if (valid) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      for (int p = p0[i]; p < p1[i]; ++p) {
        const float4 sp = sorted_s[p];
        const float dx = qp.x - sp.x, dy = qp.y - sp.y, dz = qp.z - sp.z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        n += d < r2 ? 1 : 0;
      }
  }
  sub[tid] = n;
  __syncthreads();
  // ---- block scan of the per-query totals; every slab group does it redundantly (no cross-group sync)
  int c[NSUB];
#pragma unroll
  for (int i = 0; i < NSUB; ++i) c[i] = sub[i * RQ + slot];
  const int tot = c[0] + c[1] + c[2];
  const int tot4 = (tot + 3) & ~3;  // segments start on multiples of four slots: the rank loop reads four keys per ds_read_b128
  const int inc = wave_incl_scan_add_dpp(tot4);
  const int wmx = wave_max_i32_dpp(tot);
  if (lane == WAVE - 1) wsum[tid / WAVE] = inc;
  if (lane == 0) wsum[L::THREADS / WAVE + tid / WAVE] = wmx;
  __syncthreads();
  int base = 0, total4 = 0;
#pragma unroll
  for (int i = 0; i < RQ / WAVE; ++i) {
    const int w = wsum[j * (RQ / WAVE) + i];
    if (i < slot / WAVE) base += w;
    total4 += w;
  }
  const int q_start = base + inc - tot4;
  const int my_off = q_start + (j > 0 ? c[0] : 0) + (j > 1 ? c[1] : 0);
  if (j == 0) {
    offs[slot] = q_start;
    qtot[slot] = tot;
    if (slot == RQ - 1) offs[RQ] = q_start + tot4;
  }
  int blk_flag = 0;
  const bool multi = total4 > cap;
  const bool use_rowbuf = !multi;  // rows leave through an LDS row buffer as contiguous 16-byte pieces
  const int rows_here = min(RQ, nq - blk * RQ);
  if (multi) __syncthreads();  // offs complete
  int glo = 0;
  while (glo < RQ) {
    int ghi = RQ;
    bool skip = false;
    if (multi) {
      if (tid == 0) misc[0] = RQ;
      __syncthreads();
      if (tid >= glo && tid < RQ && offs[tid + 1] - offs[glo] > cap) atomicMin(&misc[0], tid);
      __syncthreads();
      ghi = misc[0];
      if (ghi == glo) {  // one query alone overflows the key area: the caller repeats the call on the two-pass path
        blk_flag = 1;
        skip = true;
        ghi = glo + 1;
      }
    }
    const int gbase = multi ? offs[glo] : 0;
    const bool mine = valid && !skip && slot >= glo && slot < ghi;
    // ---- decode: (query slot, staged position) words of my hits into my part of my query's segment
    if (mine && j == NSUB - 1)
      for (int k = tot; k < tot4; ++k) hm[q_start - gbase + k] = PADMARK;
    if (mine && n > 0) {
      int w = my_off - gbase;
      const unsigned tag = (unsigned)slot << 16;
      if (by_mask) {
        const int c1 = 4 * nit0, c2 = 4 * (nit0 + nit1);
        const int s0 = p0[0] + rel[0], s1 = p0[1] + rel[1] - c1, s2 = p0[2] + rel[2] - c2;
        // the shift registers hold 2 bits per step: the side's first candidate sits in bit 2 S - 1 (S = steps of this thread)
        const int top = 2 * (nit0 + nit1 + nit2) - 1;
        unsigned ml = lo, mh = hi;
        while (ml | mh) {
          const int qa = 31 - __clz((int)ml), qb = 31 - __clz((int)mh);  // -1: none left on that side
          ml &= ~(qa >= 0 ? 1u << qa : 0u);
          mh &= ~(qb >= 0 ? 1u << qb : 0u);
          const int ca = 2 * (top - qa), cb = 2 * (top - qb) + 1;      // enumeration slots
          const int pa = ca + (ca < c1 ? s0 : (ca < c2 ? s1 : s2));
          const int pb = cb + (cb < c1 ? s0 : (cb < c2 ? s1 : s2));
          const int wa = qa >= 0 ? w : dummy;
          w += qa >= 0 ? 1 : 0;
          const int wb = qb >= 0 ? w : dummy;
          w += qb >= 0 ? 1 : 0;
          hm[wa] = tag | (unsigned)pa;
          hm[wb] = tag | (unsigned)pb;
        }
      } else if (staged) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          for (int p = p0[i] + rel[i]; p < p1[i] + rel[i]; ++p) {
            const float dx = qp.x - sx[p], dy = qp.y - sy[p], dz = qp.z - sz[p];
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < r2) hm[w++] = tag | (unsigned)p;
          }
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          for (int p = p0[i]; p < p1[i]; ++p) {
            const float4 sp = sorted_s[p];
            const float dx = qp.x - sp.x, dy = qp.y - sp.y, dz = qp.z - sp.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < r2) {
              hm[w] = (unsigned)p;  // position in the cell-ordered support array
              hrow[w] = (unsigned char)slot;
              ++w;
            }
          }
      }
    }
    __syncthreads();
    // ---- keys: one thread per hit slot -- distance bits and support index (balanced: no lane waits for a longer list)
    const int group_hits = skip ? 0 : offs[ghi] - gbase;
    for (int e = tid; e < group_hits; e += L::THREADS) {
      const unsigned m = hm[e];
      if (m == PADMARK) {  // larger than every real key; skipped by the ranking
        hd[e] = 0xffffffffu;
        hrow[e] = 0xff;
        continue;
      }
      float x, y, z;
      int idx, r;
      if (staged) {
        const int pp = (int)(m & 0xffffu);
        r = (int)(m >> 16);
        x = sx[pp];
        y = sy[pp];
        z = sz[pp];
        idx = si[pp];
        hrow[e] = (unsigned char)r;
      } else {
        const float4 sp = sorted_s[m];
        r = hrow[e];
        x = sp.x;
        y = sp.y;
        z = sp.z;
        idx = __float_as_int(sp.w);
      }
      const float4 qq = qbuf[r];
      const float dx = qq.x - x, dy = qq.y - y, dz = qq.z - z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      hd[e] = __float_as_uint(d);  // d >= 0: the bit pattern is monotone
      hm[e] = (unsigned)idx;
    }
    __syncthreads();
    if (use_rowbuf) {
      // the planes are dead: their place becomes the row buffer, every entry "not written"
      const int quads = (rows_here * width + 3) >> 2;
      for (int i = tid; i < quads; i += L::THREADS) reinterpret_cast<uint4*>(rowbuf)[i] = make_uint4(PADMARK, PADMARK, PADMARK, PADMARK);
      __syncthreads();
    }
    // ---- ranking: one thread per hit
    for (int e = tid; e < group_hits; e += L::THREADS) {
      const int r = hrow[e];
      if (r == 0xff) continue;
      const int a = offs[r] - gbase, quads = (offs[r + 1] - offs[r]) >> 2;
      const unsigned d = hd[e];
      const unsigned idx = hm[e];
      const uint4* seg = reinterpret_cast<const uint4*>(hd + a);
      int rank = 0;
      if (use_rowbuf) {
        int jj = 0;
        for (; jj + 4 <= quads; jj += 4) {
          uint4 k4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) k4[u] = seg[jj + u];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            rank += (k4[u].x < d ? 1 : 0) + (k4[u].y < d ? 1 : 0) + (k4[u].z < d ? 1 : 0) + (k4[u].w < d ? 1 : 0);
        }
        for (; jj < quads; ++jj) {
          const uint4 k = seg[jj];
          rank += (k.x < d ? 1 : 0) + (k.y < d ? 1 : 0) + (k.z < d ? 1 : 0) + (k.w < d ? 1 : 0);
        }
        // equal distances meet in one entry (the smallest index stays) and leave the next one unwritten
        if (rank < width) atomicMin(&rowbuf[r * width + rank], idx);
      } else {
        // direct stores: exact (distance, index) order in one go
        const uint4* segi = reinterpret_cast<const uint4*>(hm + a);
        for (int jj = 0; jj < quads; ++jj) {
          const uint4 k = seg[jj], ki = segi[jj];
          rank += (k.x < d || (k.x == d && ki.x < idx) ? 1 : 0) + (k.y < d || (k.y == d && ki.y < idx) ? 1 : 0) +
                  (k.z < d || (k.z == d && ki.z < idx) ? 1 : 0) + (k.w < d || (k.w == d && ki.w < idx) ? 1 : 0);
        }
        if (rank < width) out[(int64_t)orig[r] * width + rank] = (int64_t)idx;
      }
    }
    if (use_rowbuf) {
      __syncthreads();
      // whole rows leave as contiguous runs: consecutive lanes, consecutive 16-byte pieces of a row.  An unwritten entry
      // below the row's hit count means two hits of that row have the same distance: the row is noted and redone below
      if ((width & 1) == 0) {
        const int w2 = width >> 1, total_pairs = rows_here * w2;
        const float inv = 1.0f / (float)w2;
        for (int i = tid; i < total_pairs; i += L::THREADS) {
          int r = (int)((float)i * inv);
          r = r * w2 > i ? r - 1 : ((r + 1) * w2 <= i ? r + 1 : r);
          const int cc = (i - r * w2) * 2;
          const int cnt = qtot[r];
          const uint2 v = *reinterpret_cast<const uint2*>(rowbuf + r * width + cc);
          if ((cc < cnt && v.x == PADMARK) || (cc + 1 < cnt && v.y == PADMARK)) {
            tie_rows[r] = 1;
            misc[1] = 1;
          }
          longlong2 o;
          o.x = cc < cnt ? (long long)v.x : (long long)pad_value;
          o.y = cc + 1 < cnt ? (long long)v.y : (long long)pad_value;
          *reinterpret_cast<longlong2*>(out + (int64_t)orig[r] * width + cc) = o;
        }
      } else {
        const int total_el = rows_here * width;
        const float inv = 1.0f / (float)width;
        for (int i = tid; i < total_el; i += L::THREADS) {
          int r = (int)((float)i * inv);
          r = r * width > i ? r - 1 : ((r + 1) * width <= i ? r + 1 : r);
          const int cc = i - r * width;
          const unsigned v = rowbuf[r * width + cc];
          if (cc < qtot[r] && v == PADMARK) {
            tie_rows[r] = 1;
            misc[1] = 1;
          }
          out[(int64_t)orig[r] * width + cc] = cc < qtot[r] ? (long long)v : (long long)pad_value;
        }
      }
      __syncthreads();
      // rows with equal distances (rare): rank their hits again on (distance, index) and overwrite the row's entries
      if (misc[1]) {
        for (int r = 0; r < rows_here; ++r) {
          if (!tie_rows[r]) continue;
          const int a = offs[r], len = qtot[r];
          for (int e = tid; e < len; e += L::THREADS) {
            const unsigned d = hd[a + e], idx = hm[a + e];
            int rank = 0;
            for (int q2 = 0; q2 < len; ++q2) {
              const unsigned dk = hd[a + q2], ik = hm[a + q2];
              rank += (dk < d || (dk == d && ik < idx)) ? 1 : 0;
            }
            if (rank < width) out[(int64_t)orig[r] * width + rank] = (int64_t)idx;
          }
        }
      }
    } else {
      // ---- padding of the group's rows: half a wave per row
      for (int r = glo + tid / 32; r < min(ghi, rows_here); r += L::THREADS / 32) {
        int64_t* row = out + (int64_t)orig[r] * width;
        for (int cc = (skip ? 0 : min(qtot[r], width)) + (lane & 31); cc < width; cc += 32) row[cc] = pad_value;
      }
      if (multi) __syncthreads();  // the next group overwrites the key area
    }
    glo = ghi;
  }
  if (tid == 0) {
    int mx = 0;
#pragma unroll
    for (int i = 0; i < RQ / WAVE; ++i) mx = max(mx, wsum[L::THREADS / WAVE + i]);
    blk_stats[2 * blk] = mx;
    blk_stats[2 * blk + 1] = blk_flag;
  }
}

#include "radius_tq.hpp"

// max / max over the per-block (max hits per query, hits per block) pairs -> hdr
// mail (optional): the header also goes to the host's mailbox page, stamped (common.hpp) -- no copy, no stream synchronise
__global__ __launch_bounds__(1024) void reduce_stats_kernel(const int32_t* __restrict__ blk_stats,
                                                            int blocks, RadiusHdr* __restrict__ hdr, int32_t* mail, int stamp,
                                                            int tq) {
  // tq: the second word of a block is (give-up code | wave-finished queries << 8): max of the codes, sum of the counts
  __shared__ int sh[3][1024 / WAVE];
  int mx = 0, ms = 0, sum = 0;
  for (int i = threadIdx.x; i < blocks; i += 1024) {
    mx = max(mx, blk_stats[2 * i]);
    const int v = blk_stats[2 * i + 1];
    ms = max(ms, tq ? (v & 255) : v);
    sum += tq ? (v >> 8) : 0;
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    mx = max(mx, __shfl_xor(mx, d, WAVE));
    ms = max(ms, __shfl_xor(ms, d, WAVE));
    sum += __shfl_xor(sum, d, WAVE);
  }
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    sh[0][threadIdx.x / WAVE] = mx;
    sh[1][threadIdx.x / WAVE] = ms;
    sh[2][threadIdx.x / WAVE] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = 0, ms = 0, sum = 0;
    for (int i = 0; i < 1024 / WAVE; ++i) {
      mx = max(mx, sh[0][i]);
      ms = max(ms, sh[1][i]);
      sum += sh[2][i];
    }
    hdr->max_count = (unsigned)mx;
    hdr->max_block_hits = (unsigned)ms;
    hdr->slow_sum = sum;
    if (mail) {
      mail[0] = mx;
      mail[1] = ms;
      mail[2] = hdr->total_cells;
      mail[3] = hdr->total_sup;
      mail[4] = sum;
      mail_post(mail + 5, stamp);
    }
  }
}

__global__ void pad_fill_kernel(int64_t* __restrict__ out, int64_t n, int64_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

// Reduces the per-block statistics into the header and brings the header to the host: through the mailbox page (the
// kernel posts it, the host polls -- no copy in the stream, no stream synchronise), else by a copy into pinned memory and a
// synchronise.  Everything queued on `stream` before is complete when this returns.
inline int reduce_and_read(const RadiusWs& w, int blocks, hipStream_t stream, RadiusHdr* h_out, bool tq = false) {
  volatile int32_t* mail = mailbox();
  if (mail) mail += MAIL_RADIUS;
  const int stamp = mail ? mailbox_next_stamp() : 0;
  if (mail) mailbox_arm(mail + 5);
  hipLaunchKernelGGL(reduce_stats_kernel, dim3(1), dim3(1024), 0, stream, w.blk_stats, blocks, w.hdr,
                     const_cast<int32_t*>(mail), stamp, tq ? 1 : 0);
  GR_LAUNCH_CHECK();
  if (mail) {
    int rc = mailbox_wait(mail + 5, stamp, stream, "radius search");
    if (rc != GR_OK) return rc;
    h_out->max_count = (unsigned)mail[0];
    h_out->max_block_hits = (unsigned)mail[1];
    h_out->total_cells = mail[2];
    h_out->total_sup = mail[3];
    h_out->slow_sum = mail[4];
    return GR_OK;
  }
  RadiusHdr* h_pinned = static_cast<RadiusHdr*>(pinned_scratch(3, sizeof(RadiusHdr)));
  GR_REQUIRE(h_pinned != nullptr, "pinned read-back buffer could not be allocated");
  GR_HIP(hipMemcpyAsync(h_pinned, w.hdr, sizeof(RadiusHdr), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  *h_out = *h_pinned;
  return GR_OK;
}

template <int RQ>
int launch_count(const RadiusWs& w, const float4* sorted_q, int64_t nq, int64_t ns, int nb, const int32_t* start_s,
                 float r2, bool mono, hipStream_t stream, RadiusHdr* h_out) {  // h_out: the header, on the host when this returns
  using L = TravLds<RQ>;
  const int blocks = (int)((nq + RQ - 1) / RQ);
  const int grid = (blocks + 7) / 8 * 8;
  KernelTimer timer("radius_count", stream);
  hipLaunchKernelGGL((traverse_kernel<RQ, false, true>), dim3(grid), dim3(L::THREADS), L::count_bytes(nb <= L::TABLE_MAX ? nb : 0), stream, sorted_q,
                     (int)nq, w.q_off, nb, w.grids, start_s, w.sorted_s, r2, w.q_count, w.q_rng, w.q_mask, w.blk_stats,
                     0, 0, ns, (int64_t*)nullptr, 0, (unsigned long long*)nullptr, (unsigned char*)nullptr, mono ? 1 : 0);
  return reduce_and_read(w, blocks, stream, h_out);
}

template <int RQ>
int launch_fill(const RadiusWs& w, const float4* sorted_q, int64_t nq, int64_t ns, int nb, float r2, int64_t width,
                int64_t row_stride, int64_t max_block_hits, int64_t* out, hipStream_t stream) {
  using L = TravLds<RQ>;
  const int blocks = (int)((nq + RQ - 1) / RQ);
  const int grid = (blocks + 7) / 8 * 8;
  const int64_t cap = L::slots(max_block_hits);
  const size_t lds = L::total(cap);
  KernelTimer timer("radius_fill", stream);
  if (lds <= 160 * 1024) {
    auto kern = traverse_kernel<RQ, true, true>;
    if (lds > 64 * 1024)
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(L::THREADS), lds, stream, sorted_q, (int)nq, w.q_off, nb, w.grids,
                       w.start, w.sorted_s, r2, w.q_count, w.q_rng, w.q_mask, w.blk_stats, (int)width, (int)row_stride, ns, out,
                       (int)cap, (unsigned long long*)nullptr, (unsigned char*)nullptr, 0);
  } else {
    // very dense neighbourhoods: hit lists live in a scratch allocation owned by this call
    char* scratch = nullptr;
    const size_t per_block = (size_t)cap;
    GR_HIP(hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)grid * per_block * 9 + 256, stream));
    unsigned long long* g_hits = reinterpret_cast<unsigned long long*>(scratch);
    unsigned char* g_rows = reinterpret_cast<unsigned char*>(scratch + (size_t)grid * per_block * 8);
    hipLaunchKernelGGL((traverse_kernel<RQ, true, false>), dim3(grid), dim3(L::THREADS), L::FILL_OFF, stream,
                       sorted_q, (int)nq, w.q_off, nb, w.grids, w.start, w.sorted_s, r2, w.q_count, w.q_rng,
                       w.q_mask, w.blk_stats, (int)width, (int)row_stride, ns, out, (int)cap, g_hits, g_rows, 0);
    GR_HIP(hipFreeAsync(scratch, stream));
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

constexpr int FUSED_RQ = 64;     // queries per block: 30 KB of LDS, five blocks per CU (0.43 ms per 8 x 200 k; 128 queries: 0.50 ms)
constexpr int FUSED_PER_Q = 28;  // key slots per query in a block's key area

int launch_fused(const RadiusWs& w, const float4* sorted_q, int64_t nq, int64_t ns, int nb, const int32_t* start_s,
                 float r2, int64_t width, int64_t* out, bool mono, hipStream_t stream, RadiusHdr* h_out) {
  constexpr int RQ = FUSED_RQ;
  using L = FusedLds<RQ>;
  const int blocks = (int)((nq + RQ - 1) / RQ);
  const int grid = (blocks + 7) / 8 * 8;
  const int tcap = nb <= L::TABLE_MAX ? nb : 0;
  // key area: FUSED_PER_Q slots per query, never less than two full rows, within the 160 KB of a CU
  int cap = max(FUSED_PER_Q * RQ, (int)(2 * width + 2));
  cap = (cap + 15) / 16 * 16;
  while (L::total((int)width, cap, tcap) > 160 * 1024 && cap > 64) cap -= 16;
  const size_t region = L::region_bytes((int)width);
  const size_t lds = L::total((int)width, cap, tcap);
  GR_REQUIRE(lds <= 160 * 1024, "radius_search: neighbor_limit %lld does not fit the single-pass kernel", (long long)width);
  auto kern = fused_kernel<RQ>;
  if (lds > 64 * 1024)
    GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  {
    KernelTimer timer("radius_fused", stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(L::THREADS), lds, stream, sorted_q, (int)nq, w.q_off, nb, w.grids, start_s,
                       w.sorted_s, (int)ns, r2, w.blk_stats, (int)width, ns, out, cap, (int)region, mono ? 1 : 0);
  }
  return reduce_and_read(w, blocks, stream, h_out);
}

// gr_radius_search mode 2 / the bare search: one thread per query (radius_tq.hpp).  out != null: rows of `width` columns are
// written by the kernel; out == null: tiles + counts for launch_tq_expand.  h_out->max_block_hits != 0 = a workgroup could not
// finish: the caller repeats the call on count + fill.
int launch_tq(const RadiusWs& w, const float4* sorted_q, int64_t nq, int64_t ns, int nb, const int32_t* start_s, float r2,
              int64_t width, int64_t* out, bool mono, hipStream_t stream, RadiusHdr* h_out, int net = 32) {
  const int blocks = (int)((nq + WAVE - 1) / WAVE);
  const int grid = (blocks + 7) / 8 * 8;
  {
    KernelTimer timer("radius_tq", stream);
    const int stop = getenv("TQ_STOP") ? atoi(getenv("TQ_STOP")) : 0;
    const size_t rows_hi = (size_t)((nq + 63) / 64) * 64 * 32;
#define TQ_GO(NET)                                                                                                          \
  if (out)                                                                                                                  \
    hipLaunchKernelGGL((tq_kernel<NET, true>), dim3(grid), dim3(WAVE), 0, stream, sorted_q, (int)nq, w.q_off, nb, w.grids,   \
                       start_s, w.sorted_s, w.plane_x, w.plane_y, w.plane_z, (int)ns, r2, w.blk_stats, (int)width, ns, out, \
                       (uint32_t*)nullptr, (int32_t*)nullptr, (size_t)0, mono ? 1 : 0, stop);            \
  else                                                                                                                      \
    hipLaunchKernelGGL((tq_kernel<NET, false>), dim3(grid), dim3(WAVE), 0, stream, sorted_q, (int)nq, w.q_off, nb, w.grids,  \
                       start_s, w.sorted_s, w.plane_x, w.plane_y, w.plane_z, (int)ns, r2, w.blk_stats, 0, ns,               \
                       (int64_t*)nullptr, w.tiles, w.q_count, rows_hi, mono ? 1 : 0, 0)
    if (net == 65 && out)  // the 64-hit network behind a pre-selection of the `width` nearest hits (radius_tq.hpp, PRESEL)
      hipLaunchKernelGGL((tq_kernel<64, true, true>), dim3(grid), dim3(WAVE), 0, stream, sorted_q, (int)nq, w.q_off, nb, w.grids,
                         start_s, w.sorted_s, w.plane_x, w.plane_y, w.plane_z, (int)ns, r2, w.blk_stats, (int)width, ns, out,
                         (uint32_t*)nullptr, (int32_t*)nullptr, (size_t)0, mono ? 1 : 0, stop);
    else if (net >= 64) TQ_GO(64);
    else TQ_GO(32);
#undef TQ_GO
  }
  return reduce_and_read(w, blocks, stream, h_out, true);
}

int launch_tq_expand(const RadiusWs& w, int64_t nq, int64_t ns, int64_t width, int64_t* out, hipStream_t stream) {
  KernelTimer timer("radius_expand", stream);
  hipLaunchKernelGGL(tq_expand_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(256), 0, stream, w.tiles, w.q_count,
                     (size_t)((nq + 63) / 64) * 64 * 32, (int)nq, (int)width, ns, out);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

inline bool fused_fits(int64_t width) {
  // the row buffer / key area of the largest configuration must fit next to the candidate planes
  return width >= 1 && FusedLds<FUSED_RQ>::total((int)width, (int)((2 * width + 2 + 15) / 16 * 16), 0) <= 160 * 1024;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_radius_workspace_bytes(int64_t nq, int64_t ns, int64_t batch) {
  if (nq < 0 || ns < 0 || batch < 0) return 0;
  return carve(nullptr, nq, ns, batch).bytes;
}

extern "C" int gr_radius_count(const float* q, const float* s, const int64_t* h_q_lengths,
                               const int64_t* h_s_lengths, int64_t nq, int64_t ns, int64_t batch,
                               float radius, void* ws, size_t ws_bytes, int64_t* h_info,
                               void* stream_) {
  return gr_radius_count_cached(q, s, h_q_lengths, h_s_lengths, nq, ns, batch, radius, ws, ws_bytes, h_info, nullptr, 0,
                                stream_);
}

namespace gr {
namespace {
struct Prepared {
  RadiusWs w;
  const float4* sorted_q;
  const int32_t* start_s;
  float r2;
  int nb;
  bool same;
  bool empty;  // nothing to search: width 0
};

// Everything up to the first traversal: argument checks, offsets, support (and query) binning.  `same` = self-search.
int radius_prepare(const float* q, const float* s, const int64_t* h_q_lengths, const int64_t* h_s_lengths, int64_t nq,
                   int64_t ns, int64_t batch, float radius, void* ws, size_t ws_bytes, int64_t* h_support_sig,
                   int reuse_support, hipStream_t stream, Prepared* out_p) {
  Prepared& P = *out_p;
  P.empty = true;
  GR_REQUIRE(nq >= 0 && ns >= 0 && batch >= 0, "negative size");
  GR_REQUIRE(nq < (1ll << 31) - 1 && ns < (1ll << 31) - 1 && batch < (1 << 20),
             "radius_neighbors: sizes must fit int32 (nq=%lld ns=%lld)", (long long)nq, (long long)ns);
  int64_t sq = 0, ss = 0;
  for (int64_t b = 0; b < batch; ++b) {
    GR_REQUIRE(h_q_lengths[b] >= 0 && h_s_lengths[b] >= 0, "negative length in batch element %lld", (long long)b);
    sq += h_q_lengths[b];
    ss += h_s_lengths[b];
  }
  GR_REQUIRE(sq == nq && ss == ns, "lengths do not sum to the number of points (q %lld vs %lld, s %lld vs %lld)",
             (long long)sq, (long long)nq, (long long)ss, (long long)ns);
  if (nq == 0 || ns == 0 || batch == 0) return GR_OK;  // width 0
  RadiusWs w = carve(ws, nq, ns, batch);
  if (ws == nullptr || ws_bytes < w.bytes) {
    set_error("radius workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return GR_ERR_WORKSPACE;
  }
  const bool same = (q == s) && (nq == ns) && memcmp(h_q_lengths, h_s_lengths, sizeof(int64_t) * batch) == 0;
  // signature of the support side (cloud pointer, sizes, radius, lengths): lets a caller that searches the same
  // supports again (other queries, same radius -- the three searches per level of the data pyramid) skip the binning
  int64_t sig[4] = {ns, batch, 0, (int64_t)reinterpret_cast<uintptr_t>(s)};
  {
    uint32_t rb;
    memcpy(&rb, &radius, 4);
    uint64_t hsh = 1469598103934665603ull ^ rb;
    for (int64_t b = 0; b < batch; ++b) hsh = (hsh ^ (uint64_t)h_s_lengths[b]) * 1099511628211ull;
    sig[2] = (int64_t)hsh;
  }
  const bool reuse = reuse_support != 0;
  if (reuse) {
    GR_REQUIRE(h_support_sig != nullptr, "reuse_support needs the signature written by the preparing call");
    GR_REQUIRE(memcmp(sig, h_support_sig, sizeof(sig)) == 0,
               "reuse_support: supports / lengths / radius differ from the call that prepared this workspace");
  }
  if (h_support_sig) memcpy(h_support_sig, sig, sizeof(sig));
  // offsets (host -> device)
  // q offsets | s offsets | bbox block offsets, staged in pinned memory (pinned_scratch: every call ends with a stream
  // synchronise, so the previous call's copy has left the buffer)
  int32_t* const h_offsets = static_cast<int32_t*>(pinned_scratch(2, sizeof(int32_t) * 3 * (batch + 1)));
  GR_REQUIRE(h_offsets != nullptr, "pinned staging buffer could not be allocated");
  {
    int32_t* tmp = h_offsets;
    tmp[0] = 0;
    tmp[batch + 1] = 0;
    for (int64_t b = 0; b < batch; ++b) {
      tmp[b + 1] = tmp[b] + (int32_t)h_q_lengths[b];
      tmp[batch + 1 + b + 1] = tmp[batch + 1 + b] + (int32_t)h_s_lengths[b];
    }
    if (!reuse) {  // first bounding-box block of every cloud
      int32_t* blk = tmp + 2 * (batch + 1);
      blk[0] = 0;
      for (int64_t b = 0; b < batch; ++b) blk[b + 1] = blk[b] + (int32_t)((h_s_lengths[b] + BBOX_PTS - 1) / BBOX_PTS);
    }
    // (few clouds, full binning: the offsets ride in the first launch's arguments instead -- see OffsetArgs)
    if (reuse || batch > KARG_CLOUDS)
      GR_HIP(hipMemcpyAsync(w.q_off, tmp, sizeof(int32_t) * (reuse ? 1 : 3) * (batch + 1), hipMemcpyHostToDevice, stream));
  }
  const int nb = (int)batch;
  int32_t* start_s = w.start;
  KernelTimer bin_timer("radius_bin", stream);  // bbox .. cell order (nothing is launched when the grid is reused in a self-search)
  const int64_t su = w.nsup + 1;
  BinSide A{s, (int)ns, w.s_off, w.s_cell, w.pairs_s, w.sup_zero, w.sup_zero + 2 * su, w.sup_start, w.sorted_s, start_s,
            w.plane_x, w.plane_y, w.plane_z};
  BinSide B{q, (int)nq, w.q_off, w.q_cell, w.pairs_q, w.sup_zero + su, w.sup_zero + 3 * su, w.sup_start + su, w.sorted_q, nullptr,
            nullptr, nullptr, nullptr};
  const int blocks_s = (int)((ns + COARSE_PTS - 1) / COARSE_PTS), blocks_q = (int)((nq + COARSE_PTS - 1) / COARSE_PTS);
  if (!reuse) {
    // ---- supports (and, in the same launches, the queries): bbox, grid, two-level counting sort
    {
      const int nzero = (int)(4 * su);
      const int bbox_blocks = h_offsets[2 * (batch + 1) + batch];
      OffsetArgs ka;
      if (batch <= KARG_CLOUDS) {
        GR_REQUIRE(w.s_off == w.q_off + (batch + 1) && w.blk_off == w.q_off + 2 * (batch + 1), "radius workspace layout");
        memcpy(ka.v, h_offsets, sizeof(int32_t) * 3 * (batch + 1));
        hipLaunchKernelGGL((bbox_partial_kernel<true>), dim3(bbox_blocks), dim3(256), 0, stream, s, w.s_off, w.blk_off, nb,
                           w.bbox_partial, w.sup_zero, nzero, ka, w.q_off);
      } else {
        hipLaunchKernelGGL((bbox_partial_kernel<false>), dim3(bbox_blocks), dim3(256), 0, stream, s, w.s_off, w.blk_off, nb,
                           w.bbox_partial, w.sup_zero, nzero, ka, w.q_off);
      }
    }
    // x sub-cells per cell: 2 measured best end to end (count pass 0.166 -> 0.157 ms; 8 gives 0.150 ms but the scan and the
    // scatter over an 8x larger cell table take the difference back)
    constexpr int xk_max = 2;
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(256), 0, stream, w.bbox, w.bbox_partial, w.blk_off, w.s_off, nb, radius,
                       xk_max, w.grids, w.hdr, w.sup_off);
    const int bq = same ? 0 : blocks_q;
    hipLaunchKernelGGL((coarse_kernel<false>), dim3(blocks_s + bq), dim3(256), 0, stream, A, B, blocks_s, nb, w.grids);
    hipLaunchKernelGGL(sup_scan_kernel, dim3(same ? 1 : 2), dim3(1024), 0, stream, A, B, 0, w.hdr);
    hipLaunchKernelGGL((coarse_kernel<true>), dim3(blocks_s + bq), dim3(256), 0, stream, A, B, blocks_s, nb, w.grids);
    const int fine_blocks = (int)std::min<int64_t>(w.nsup, 4096);
    hipLaunchKernelGGL(fine_kernel, dim3((unsigned)(fine_blocks * (same ? 1 : 2))), dim3(256), 0, stream, A, B, fine_blocks, nb, w.grids,
                       w.sup_off, w.hdr);
    GR_LAUNCH_CHECK();
  } else if (!same) {
    // ---- the support grid is in place: only the queries are binned into it
    const int nzero = (int)su;
    hipLaunchKernelGGL(bin_init2_kernel, dim3(std::min(256, (nzero + 255) / 256)), dim3(256), 0, stream, w.sup_zero + su,
                       w.sup_zero + 3 * su, nzero);
    hipLaunchKernelGGL((coarse_kernel<false>), dim3(blocks_q), dim3(256), 0, stream, A, B, 0, nb, w.grids);
    hipLaunchKernelGGL(sup_scan_kernel, dim3(1), dim3(1024), 0, stream, A, B, 1, w.hdr);
    hipLaunchKernelGGL((coarse_kernel<true>), dim3(blocks_q), dim3(256), 0, stream, A, B, 0, nb, w.grids);
    hipLaunchKernelGGL(fine_kernel, dim3((unsigned)std::min<int64_t>(w.nsup, 4096)), dim3(256), 0, stream, A, B, 0, nb, w.grids, w.sup_off,
                       w.hdr);
    GR_LAUNCH_CHECK();
  }
  P.w = w;
  P.sorted_q = same ? w.sorted_s : w.sorted_q;
  P.start_s = start_s;
  P.r2 = radius * radius;  // radius_neighbors_cpu.cpp:12 (fp32 product)
  P.nb = nb;
  P.same = same;
  P.empty = false;
  return GR_OK;
}
}  // namespace
}  // namespace gr

namespace gr {
namespace {
// 0 = count, host, fill; 1 = the single-pass kernel (three threads per query); 2 = one thread per query (radius_tq.hpp),
// always tried first (32-hit network); 3 (default) = the kernel tq_choice picks for the call site (32-hit network, 64-hit network,
// 64-hit network behind the pre-selection, or count + fill); 4 = the 64-hit network always tried first; 5 = the same behind
// the pre-selection (limited searches; the bare search has no width to select for and takes the plain 64-hit network).
// Initialised from GR_RADIUS_SINGLE_PASS.
std::atomic<int>& search_mode() {
  static std::atomic<int> mode{[] {
    const char* a = getenv("GR_RADIUS_SINGLE_PASS");
    return (a && a[0] >= '0' && a[0] <= '5') ? a[0] - '0' : 3;
  }()};
  return mode;
}
}  // namespace
}  // namespace gr

namespace gr {
namespace {
// Which search kernel a call site gets.  The thread-per-query kernel sorts up to NET hits per query in registers (NET = 32:
// the big levels of the data pyramid, 4 - 14 hits on average; NET = 64: its middle levels, ~30); where most queries of a
// wave have more, the kernel gives up after its tests and the call is repeated on count + fill.  A caller repeats the same
// (radius, limit) call site over and over (13 per pair in the pyramid), so a give-up is remembered per (radius bits,
// limit): the site moves 32 -> 64 -> 64 behind the pre-selection (rows of a known width <= TQ_PRESEL_MAX: the coarsest levels,
// where a query has ~150 hits and keeps 49) -> count + fill, and steps back down one level every TQ_RETRY_AFTER calls.
constexpr int TQ_MEMO = 64, TQ_RETRY_AFTER = 256;
constexpr int64_t TQ_PRESEL_MAX = 56;  // the selection needs a bin boundary between `width` and 64 hits
struct TqMemo {
  uint32_t rbits;
  int64_t limit;
  int level;  // 0: the 32-hit network, 1: the 64-hit network, 2: the 64-hit network behind the pre-selection, 3: count + fill
  int calls;  // calls at this level since it last changed (a site at level > 0 steps back down every TQ_RETRY_AFTER calls)
  bool used;
};
TqMemo g_tq_memo[TQ_MEMO];
std::mutex g_tq_memo_mu;

TqMemo* tq_find(uint32_t rb, int64_t limit) {
  for (TqMemo& e : g_tq_memo)
    if (e.used && e.rbits == rb && e.limit == limit) return &e;
  return nullptr;
}

inline bool tq_presel_ok(int64_t limit) { return limit >= 1 && limit <= TQ_PRESEL_MAX; }

// which kernel this call site gets: 32 / 64 = the thread-per-query kernel with that network, 65 = the 64-hit network behind the
// pre-selection, 0 = count + fill
int tq_choice(float radius, int64_t limit) {
  uint32_t rb;
  memcpy(&rb, &radius, 4);
  std::lock_guard<std::mutex> lk(g_tq_memo_mu);
  TqMemo* e = tq_find(rb, limit);
  if (!e) return 32;
  if (e->level > 0 && ++e->calls > TQ_RETRY_AFTER) {
    e->level -= (e->level == 3 && !tq_presel_ok(limit)) ? 2 : 1;
    e->calls = 0;
  }
  return e->level == 0 ? 32 : (e->level == 1 ? 64 : (e->level == 2 ? 65 : 0));
}

void tq_report(float radius, int64_t limit, int net, bool gave_up) {
  if (!gave_up) return;
  uint32_t rb;
  memcpy(&rb, &radius, 4);
  std::lock_guard<std::mutex> lk(g_tq_memo_mu);
  TqMemo* e = tq_find(rb, limit);
  if (!e) {
    static int next = 0;
    for (TqMemo& c : g_tq_memo)
      if (!c.used && !e) e = &c;
    if (!e) e = &g_tq_memo[next++ % TQ_MEMO];
    e->used = true;
    e->rbits = rb;
    e->limit = limit;
    e->level = 0;
  }
  e->level = net == 32 ? 1 : (net == 64 && tq_presel_ok(limit) ? 2 : 3);
  e->calls = 0;
}
}  // namespace
}  // namespace gr

extern "C" int gr_radius_count_cached(const float* q, const float* s, const int64_t* h_q_lengths,
                                      const int64_t* h_s_lengths, int64_t nq, int64_t ns, int64_t batch,
                                      float radius, void* ws, size_t ws_bytes, int64_t* h_info,
                                      int64_t* h_support_sig, int reuse_support, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_info != nullptr, "h_info is null");
  h_info[0] = h_info[1] = h_info[2] = h_info[3] = 0;
  Prepared P;
  int rc = radius_prepare(q, s, h_q_lengths, h_s_lengths, nq, ns, batch, radius, ws, ws_bytes, h_support_sig, reuse_support,
                          stream, &P);
  if (rc != GR_OK) return rc;
  if (P.empty) return GR_OK;  // width 0
  const RadiusWs& w = P.w;
  const bool same = P.same;
  RadiusHdr h;
  const int mode = search_mode().load();
  const int net = mode == 2 ? 32 : (mode == 4 || mode == 5 ? 64 : (mode == 3 && ns < (1ll << 29) ? tq_choice(radius, -1) : 0));
  if (net != 0 && ns < (1ll << 29)) {
    // one thread per query: the whole search now (sorted compact rows), gr_radius_fill only widens them
    rc = launch_tq(w, P.sorted_q, nq, ns, P.nb, P.start_s, P.r2, 0, nullptr, same, stream, &h, net);
    if (rc != GR_OK) return rc;
    const bool done = h.max_block_hits == 0 && h.max_count <= (unsigned)TQ_ROW_CAP;
    // (a finished call most of whose waves needed the exact path is reported too: the next call of the site starts higher)
    if (mode == 3) tq_report(radius, -1, net, !done || (int64_t)h.slow_sum * 8 > nq);
    if (done) {
      h_info[0] = h.max_count;
      h_info[1] = -1;  // the tiles are in the workspace
      h_info[2] = same ? 1 : 0;
      h_info[3] = h.total_cells;
      return GR_OK;
    }
  }
  rc = launch_count<RT>(w, P.sorted_q, nq, ns, P.nb, P.start_s, P.r2, same, stream, &h);
  if (rc != GR_OK) return rc;
  h_info[0] = h.max_count;
  h_info[1] = h.max_block_hits;
  h_info[2] = same ? 1 : 0;
  h_info[3] = h.total_cells;
  return GR_OK;
}

extern "C" int gr_radius_fill(const float* q, const float* s, int64_t nq, int64_t ns, int64_t batch,
                              float radius, int64_t width, const int64_t* h_info, int64_t* out,
                              void* ws, size_t ws_bytes, void* stream_) {
  (void)q;
  (void)s;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_info != nullptr, "h_info is null");
  GR_REQUIRE(width >= 0 && width <= (1 << 30), "bad width %lld", (long long)width);
  if (nq == 0 || width == 0) return GR_OK;
  GR_REQUIRE(out != nullptr, "out is null");
  if (ns == 0 || batch == 0 || h_info[0] == 0) {
    // nothing matched anywhere: a caller that insists on a fixed width gets all-padding rows
    const int64_t n = nq * width;
    hipLaunchKernelGGL(pad_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, ns);
    GR_LAUNCH_CHECK();
    return GR_OK;
  }
  RadiusWs w = carve(ws, nq, ns, batch);
  if (ws == nullptr || ws_bytes < w.bytes) {
    set_error("radius workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return GR_ERR_WORKSPACE;
  }
  const bool same = h_info[2] != 0;
  const float4* sorted_q = same ? w.sorted_s : w.sorted_q;
  const float r2 = radius * radius;
  if (h_info[1] == -1) return launch_tq_expand(w, nq, ns, width, out, stream);
  return launch_fill<RT>(w, sorted_q, nq, ns, (int)batch, r2, width, width, h_info[1], out, stream);
}

extern "C" int gr_radius_search_mode(int mode) {
  const int old = search_mode().load();
  if (mode >= 0 && mode <= 5) search_mode().store(mode);
  return old;
}

extern "C" int gr_radius_search(const float* q, const float* s, const int64_t* h_q_lengths, const int64_t* h_s_lengths,
                                int64_t nq, int64_t ns, int64_t batch, float radius, int64_t limit, int64_t* out,
                                void* ws, size_t ws_bytes, int64_t* h_info, int64_t* h_support_sig, int reuse_support,
                                void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_info != nullptr, "h_info is null");
  for (int i = 0; i < 6; ++i) h_info[i] = 0;
  GR_REQUIRE(limit >= 1 && limit <= (1 << 20), "radius_search: neighbor_limit must be positive (got %lld)", (long long)limit);
  GR_REQUIRE(out != nullptr || nq == 0, "out is null");
  Prepared P;
  int rc = radius_prepare(q, s, h_q_lengths, h_s_lengths, nq, ns, batch, radius, ws, ws_bytes, h_support_sig, reuse_support,
                          stream, &P);
  if (rc != GR_OK) return rc;
  if (P.empty) return GR_OK;  // width 0
  const RadiusWs& w = P.w;
  // GR_RADIUS_SINGLE_PASS=1 selects the single-pass kernel (fused_kernel above).  It is not the default: on 8 x 200 k points it
  // runs as long as count + fill together (both are bound by VALU issue: ~3 000 instructions per wave either way)
  const int mode = search_mode().load();
  const int net = ns >= (1ll << 29) ? 0 : (mode == 2 ? 32 : (mode == 4 ? 64 : (mode == 5 ? 65 : (mode == 3 ? tq_choice(radius, limit) : 0))));
  const bool tq = net != 0;
  const bool fused = (mode == 1 && fused_fits(limit)) || tq;
  if (fused) {
    RadiusHdr hf;
    rc = tq ? launch_tq(w, P.sorted_q, nq, ns, P.nb, P.start_s, P.r2, limit, out, P.same, stream, &hf, net)
            : launch_fused(w, P.sorted_q, nq, ns, P.nb, P.start_s, P.r2, limit, out, P.same, stream, &hf);
    if (rc != GR_OK) return rc;
    if (tq && mode == 3) tq_report(radius, limit, net, hf.max_block_hits != 0 || (int64_t)hf.slow_sum * 8 > nq);
    h_info[0] = hf.max_count;
    h_info[2] = P.same ? 1 : 0;
    h_info[3] = hf.total_cells;
    if (hf.max_block_hits == 0) {  // no query overflowed its block's key area: `out` is complete
      h_info[4] = 1;
      return GR_OK;
    }
  }
  // count, host, fill: the first min(max_count, limit) columns of the (nq, limit) rows.  (Measured and dropped: launching the
  // fill behind the count with an LDS key area sized from the previous call of the same shape, to take the host out of the
  // middle -- 0.594 vs 0.579 ms per 8 x 200 k points: the host prepares the fill while the count runs; what is left of it
  // sits between calls, not between the kernels.)
  RadiusHdr h;
  rc = launch_count<RT>(w, P.sorted_q, nq, ns, P.nb, P.start_s, P.r2, P.same, stream, &h);
  if (rc != GR_OK) return rc;
  h_info[0] = h.max_count;
  h_info[1] = h.max_block_hits;
  h_info[2] = P.same ? 1 : 0;
  h_info[3] = h.total_cells;
  h_info[4] = 0;
  if (h.max_count == 0) return GR_OK;  // width 0
  const int64_t width = h.max_count < (uint64_t)limit ? (int64_t)h.max_count : limit;
  return launch_fill<RT>(w, P.sorted_q, nq, ns, P.nb, P.r2, width, limit, h.max_block_hits, out, stream);
}
