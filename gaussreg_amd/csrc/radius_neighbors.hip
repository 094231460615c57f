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
//   bbox        per-cloud bounding box of the supports (block reductions, six atomics per block)
//   grid_setup  per-cloud cell edge (>= radius, coarsened so cells <= max(4096, 4 n_b)), dims, bases
//   bin_count   cell id per support / per query; the histogram atomic returns the point's slot inside its cell
//   scan        exclusive scan of the histograms, stopped at the real cell count (common.hip)
//   scatter     counting-sort supports and queries into cell order as float4 {x,y,z,orig index} (no atomics)
//   count       thread per (cell-ordered query, z-slab): candidates staged in LDS as coordinate planes, tested two at
//               a time with packed fp32 math; per thread a hit count, the nine candidate ranges and a hit bit mask;
//               max over queries -> host (the row width the reference returns)
//   fill        gathers only the hits named by the masks into per-query LDS segments, ranks every hit inside its
//               segment by counting (one thread per hit) and stores it at out[query][rank]; pads the rows
// gr_radius_count_cached lets consecutive searches over the same supports and radius skip bbox .. scatter for the
// support side (the data pyramid searches every level's supports three times).
#include <vector>

#include "common.hpp"

namespace gr {
namespace {

struct BatchGrid {  // 64 bytes: copied to LDS as four int4
  double org[3];
  double inv_cell;    // y and z: cells of edge >= r (1 + 2^-10)
  double inv_cell_x;  // x: `xk` sub-cells per cell -- the x window of a query shrinks from 3 r to (2 + 1/xk) r while every
                      // (y, z) row of cells stays one contiguous range of the cell-sorted supports (x is the fastest index)
  int dim[3];         // dim[0] counts the fine x cells
  int cell_base;
  int xk;
  int pad;
};
static_assert(sizeof(BatchGrid) == 64, "BatchGrid is staged in LDS as four int4");

struct RadiusHdr {
  unsigned int max_count;
  unsigned int max_block_hits;
  int total_cells;
  int pad;
};

constexpr int RT = 128;  // queries per block in count/fill (3 threads per query)

struct RadiusWs {
  RadiusHdr* hdr;
  int32_t* q_off;
  int32_t* s_off;
  uint32_t* bbox;
  int32_t* blk_off;
  BatchGrid* grids;
  int32_t* s_cell;
  int32_t* q_cell;
  int32_t* s_rank;  // arrival order of a point inside its cell (from the histogram atomics)
  int32_t* q_rank;
  int32_t* cnt;    // [2][ccap+1]
  int32_t* start;  // [2][ccap+1]
  int32_t* scan_ws;
  float4* sorted_s;
  float4* sorted_q;
  int32_t* q_count;    // [3][nq] hits per (z-slab, query)
  int2* q_rng;         // [3 dy][3 slab][nq] candidate range (p0, p1) per band
  unsigned long long* q_mask;  // [3 slab][nq] hit bits in candidate enumeration order
  int32_t* blk_stats;  // [blocks][2]
  int64_t ccap;
  size_t bytes;
};

RadiusWs carve(void* ws, int64_t nq, int64_t ns, int64_t batch) {
  RadiusWs w;
  Carver c(ws);
  w.ccap = 4096 * batch + 4 * ns;
  w.hdr = c.take<RadiusHdr>(1);
  w.q_off = c.take<int32_t>(3 * (batch + 1));  // q offsets | s offsets | bbox block offsets: one host-to-device copy
  w.s_off = w.q_off + (batch + 1);
  w.blk_off = w.s_off + (batch + 1);
  w.bbox = c.take<uint32_t>(batch * 6);
  w.grids = c.take<BatchGrid>(batch);
  // support side first (sizes depend on ns and batch only): a later call with other queries finds it in place
  w.s_cell = c.take<int32_t>(ns);
  w.s_rank = c.take<int32_t>(ns);
  w.cnt = c.take<int32_t>(2 * (w.ccap + 1));
  w.start = c.take<int32_t>(2 * (w.ccap + 1));
  w.scan_ws = c.take<int32_t>(2 * scan_ws_ints(w.ccap + 1));
  w.sorted_s = c.take<float4>(ns);
  // query side
  w.q_cell = c.take<int32_t>(nq);
  w.q_rank = c.take<int32_t>(nq);
  w.sorted_q = c.take<float4>(nq);
  w.q_count = c.take<int32_t>(3 * nq);
  w.q_rng = c.take<int2>(9 * nq);
  w.q_mask = c.take<unsigned long long>(3 * nq);
  w.blk_stats = c.take<int32_t>(2 * ((nq + RT - 1) / RT + 1));
  w.bytes = c.used();
  return w;
}

// ---------------------------------------------------------------- grid setup
__global__ void grid_setup_kernel(const uint32_t* __restrict__ bbox,
                                  const int32_t* __restrict__ s_off, int nb, float radius, int xk_max,
                                  BatchGrid* __restrict__ grids, RadiusHdr* __restrict__ hdr) {
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
    g.pad = 0;
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
    grids[b] = g;
  }
  // exclusive prefix of the per-cloud cell counts: chunks of blockDim clouds, running carry in LDS
  // (a serial loop over global memory cost ~0.2 us per cloud)
  __shared__ int s_cnt[256];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
    const int b = b0 + threadIdx.x;
    const int cells = b < nb ? grids[b].dim[0] * grids[b].dim[1] * grids[b].dim[2] : 0;
    s_cnt[threadIdx.x] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = s_carry;
      for (int k = 0; k < (int)blockDim.x; ++k) {
        const int c = s_cnt[k];
        s_cnt[k] = acc;
        acc += c;
      }
      s_carry = acc;
    }
    __syncthreads();
    if (b < nb) grids[b].cell_base = s_cnt[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hdr->total_cells = s_carry;
    hdr->max_count = 0;
    hdr->max_block_hits = 0;
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

// ---------------------------------------------------------------- bin + histogram
__global__ __launch_bounds__(256) void bin_count_kernel(
    const float* __restrict__ s, int ns, const float* __restrict__ q, int nq,
    const int32_t* __restrict__ s_off, const int32_t* __restrict__ q_off, int nb,
    const BatchGrid* __restrict__ grids, int32_t* __restrict__ s_cell, int32_t* __restrict__ q_cell,
    int32_t* __restrict__ s_rank, int32_t* __restrict__ q_rank, int32_t* __restrict__ cnt_s,
    int32_t* __restrict__ cnt_q) {
  // a block's 256 consecutive points belong to one cloud or two neighbours almost always: wave-uniform cloud lookup for
  // the block's first and last point, and only blocks that straddle more clouds search per thread
  const int i0 = blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
  const bool is_s = i0 < ns;  // blocks never mix supports and queries unless ns is not a multiple of 256
  __shared__ int s_lohi[2];
  if (threadIdx.x == 0) {
    const int last = min(i0 + (int)blockDim.x, is_s ? ns : ns + nq) - 1;
    s_lohi[0] = is_s ? find_batch(s_off, nb, i0) : find_batch(q_off, nb, i0 - ns);
    s_lohi[1] = is_s ? find_batch(s_off, nb, last) : find_batch(q_off, nb, last - ns);
  }
  __syncthreads();
  const int blo = s_lohi[0], bhi = s_lohi[1];
  if (i < ns) {
    int b = blo;
    if (is_s && bhi != blo) b = bhi == blo + 1 ? (i >= s_off[bhi] ? bhi : blo) : find_batch(s_off, nb, i);
    if (!is_s) b = find_batch(s_off, nb, i);
    const int c = clamped_cell(grids[b], s[3 * (int64_t)i], s[3 * (int64_t)i + 1], s[3 * (int64_t)i + 2]);
    s_cell[i] = c;
    s_rank[i] = atomicAdd(&cnt_s[c], 1);  // the returned count doubles as the slot inside the cell
  } else if (i < ns + nq) {
    const int j = i - ns;
    int b = blo;
    if (is_s) b = find_batch(q_off, nb, j);  // the one block that holds the last supports and the first queries
    else if (bhi != blo) b = bhi == blo + 1 ? (j >= q_off[bhi] ? bhi : blo) : find_batch(q_off, nb, j);
    const int c = clamped_cell(grids[b], q[3 * (int64_t)j], q[3 * (int64_t)j + 1], q[3 * (int64_t)j + 2]);
    q_cell[j] = c;
    q_rank[j] = atomicAdd(&cnt_q[c], 1);
  }
}

__global__ __launch_bounds__(256) void scatter_kernel(
    const float* __restrict__ s, int ns, const float* __restrict__ q, int nq,
    const int32_t* __restrict__ s_cell, const int32_t* __restrict__ q_cell,
    const int32_t* __restrict__ start_s, const int32_t* __restrict__ start_q,
    const int32_t* __restrict__ s_rank, const int32_t* __restrict__ q_rank, float4* __restrict__ sorted_s,
    float4* __restrict__ sorted_q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) {
    const int c = s_cell[i];
    const int slot = start_s[c] + s_rank[i];
    sorted_s[slot] = make_float4(s[3 * (int64_t)i], s[3 * (int64_t)i + 1], s[3 * (int64_t)i + 2],
                                 __int_as_float(i));
  } else if (i < ns + nq) {
    const int j = i - ns;
    const int c = q_cell[j];
    const int slot = start_q[c] + q_rank[j];
    sorted_q[slot] = make_float4(q[3 * (int64_t)j], q[3 * (int64_t)j + 1], q[3 * (int64_t)j + 2],
                                 __int_as_float(j));
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
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int f = tid + u * L::THREADS;
        if (f < total) {
          int src = bl[0] + f;
#pragma unroll
          for (int k = 1; k < NBAND; ++k) src = f >= bs[k] ? bl[k] + (f - bs[k]) : src;
          v[u] = sorted_s[src];
        }
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
// caller can allocate (nq, limit) rows BEFORE anything is counted and one kernel does the whole search: set-up, staging
// and candidate tests as in the COUNT pass above (hit masks stay in registers), a block scan of the hit counts, the hits
// re-read from the staged candidates (LDS, not a second trip to memory) into per-query key segments, ranking by counting
// and the row stores.  Nothing per query goes through global memory in between (the two-pass path writes and re-reads
// 180 bytes of ranges / masks / counts per query) and the host does not sit between two launches.
//   blk_stats[2 blk]     = largest hit count of a query in the block   (max -> the width the reference would return)
//   blk_stats[2 blk + 1] = 1 if a single query had more hits than the block's key area holds (the caller then repeats
//                          the search on the two-pass path; never seen below ~3 500 neighbours per query)
// A block whose hits do not fit its key area at once works through its queries in groups.
template <int RQ>
struct FusedLds {
  static constexpr int THREADS = NSUB * RQ;
  static constexpr int STAGE_CAP = 12 * RQ;
  static constexpr int TABLE_MAX = 256;
  // ints: offs[RQ+1], orig[RQ], qtot[RQ], wsum[2 * THREADS/64], sub[3*RQ], band_lo[9], band_hi[9], band_base[10], misc[4]
  static constexpr int N_INTS = (RQ + 1) + RQ + RQ + 2 * (THREADS / WAVE) + NSUB * RQ + 9 + 9 + 10 + 4;
  static constexpr size_t STAGE_OFF = (size_t)(N_INTS * 4 + 15) / 16 * 16;
  static size_t region_bytes(int width) {  // candidate planes x, y, z, index; the row buffer takes their place later
    const size_t st = (size_t)STAGE_CAP * 16, rb = ((size_t)RQ * width * 4 + 15) / 16 * 16;
    return st > rb ? st : rb;
  }
  static size_t tables_bytes(int tcap) { return tcap > 0 ? ((size_t)(tcap + 1) * 4 + 15) / 16 * 16 + (size_t)tcap * sizeof(BatchGrid) : 0; }
  static size_t total(int width, int cap, int tcap) {
    const size_t hits = (size_t)cap * 9, tb = tables_bytes(tcap);
    return STAGE_OFF + region_bytes(width) + (hits > tb ? hits : tb);
  }
};

template <int RQ, bool ROWBUF>
__global__ __launch_bounds__(NSUB* RQ) void fused_kernel(
    const float4* __restrict__ sorted_q, int nq, const int32_t* __restrict__ q_off, int nb,
    const BatchGrid* __restrict__ grids, const int32_t* __restrict__ start_s, const float4* __restrict__ sorted_s, float r2,
    int32_t* __restrict__ blk_stats, int width, int64_t pad_value, int64_t* __restrict__ out, int cap, int region_bytes,
    int mono) {
  using L = FusedLds<RQ>;
  static_assert(RQ % WAVE == 0 && RQ <= 256, "row ids are bytes; waves must not straddle slabs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* offs = reinterpret_cast<int*>(smem);
  int* orig = offs + (RQ + 1);
  int* qtot = orig + RQ;
  int* wsum = qtot + RQ;
  int* sub = wsum + 2 * (L::THREADS / WAVE);  // [NSUB][RQ]
  int* band_lo = sub + NSUB * RQ;
  int* band_hi = band_lo + NBAND;
  int* band_base = band_hi + NBAND;
  int* misc = band_base + NBAND + 1;
  float* sx = reinterpret_cast<float*>(smem + L::STAGE_OFF);
  float* sy = sx + L::STAGE_CAP;
  float* sz = sy + L::STAGE_CAP;
  int* si = reinterpret_cast<int*>(sz + L::STAGE_CAP);
  unsigned int* rowbuf = reinterpret_cast<unsigned int*>(smem + L::STAGE_OFF);  // takes the planes' place after the emission
  char* hreg = smem + L::STAGE_OFF + region_bytes;
  unsigned long long* hits = reinterpret_cast<unsigned long long*>(hreg);
  unsigned char* rows = reinterpret_cast<unsigned char*>(hreg + (size_t)cap * 8);
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
    if (j == 0) orig[slot] = __float_as_int(qp.w);
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
    const int total = band_base[NBAND];
    int bl[NBAND], bs[NBAND];
#pragma unroll
    for (int k = 0; k < NBAND; ++k) {
      bl[k] = band_lo[k];
      bs[k] = band_base[k];
    }
    constexpr int PER = L::STAGE_CAP / L::THREADS;
    float4 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int f = tid + u * L::THREADS;
      if (f < total) {
        int src = bl[0] + f;
#pragma unroll
        for (int k = 1; k < NBAND; ++k) src = f >= bs[k] ? bl[k] + (f - bs[k]) : src;
        v[u] = sorted_s[src];
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int f = tid + u * L::THREADS;
      if (f < total) {
        sx[f] = v[u].x;
        sy[f] = v[u].y;
        sz[f] = v[u].z;
        si[f] = __float_as_int(v[u].w);
      }
    }
    __syncthreads();
  }
  // ---- test every candidate; hits are remembered as a bit mask in enumeration order (band 0, 1, 2)
  unsigned long long mask = 0ull;
  int n = 0;
  int rel[3] = {0, 0, 0};
  if (staged) {
#pragma unroll
    for (int i = 0; i < 3; ++i) rel[i] = band_base[3 * j + i] - band_lo[3 * j + i];
  }
  if (valid && staged) {
    int bitpos = 0;
    const f32x2 qx = {qp.x, qp.x}, qy = {qp.y, qp.y}, qz = {qp.z, qp.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int p = p0[i] + rel[i];
      const int e = p1[i] + rel[i];
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
  const int tot2 = (tot + 1) & ~1;  // segments start on even slots: the rank loop reads two keys per ds_read_b128
  const int inc = wave_incl_scan_add_dpp(tot2);
  const int wmx = wave_max_i32_dpp(tot);
  if (lane == WAVE - 1) wsum[tid / WAVE] = inc;
  if (lane == 0) wsum[L::THREADS / WAVE + tid / WAVE] = wmx;
  __syncthreads();
  int base = 0, total2 = 0;
#pragma unroll
  for (int i = 0; i < RQ / WAVE; ++i) {
    const int w = wsum[j * (RQ / WAVE) + i];
    if (i < slot / WAVE) base += w;
    total2 += w;
  }
  const int q_start = base + inc - tot2;
  const int my_off = q_start + (j > 0 ? c[0] : 0) + (j > 1 ? c[1] : 0);
  if (j == 0) {
    offs[slot] = q_start;
    qtot[slot] = tot;
    if (slot == RQ - 1) offs[RQ] = q_start + tot2;
  }
  int blk_flag = 0;
  const bool multi = total2 > cap;
  const bool use_rowbuf = ROWBUF && !multi;
  const int rows_here = min(RQ, nq - blk * RQ);
  const int len0 = p1[0] - p0[0], len1 = p1[1] - p0[1], len2 = p1[2] - p0[2];
  const bool by_mask = staged && (len0 + len1 + len2 <= 64);
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
    // ---- emission: (distance, index) keys of my hits into my query's segment
    if (valid && !skip && slot >= glo && slot < ghi && n > 0) {
      int w = my_off - gbase;
      if (tot2 != tot && j == NSUB - 1) {  // pad slot: larger than every real key, skipped by the rank phase
        hits[q_start - gbase + tot] = ~0ull;
        rows[q_start - gbase + tot] = 0xff;
      }
      auto emit = [&](float x, float y, float z, int idx) {
        const float dx = qp.x - x, dy = qp.y - y, dz = qp.z - z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < r2) {
          hits[w] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)idx;  // d >= 0: bit pattern is monotone
          rows[w] = (unsigned char)slot;
          ++w;
        }
      };
      if (by_mask) {
        unsigned long long bits = mask;
        const int s0 = p0[0] + rel[0], s1 = p0[1] + rel[1] - len0, s2 = p0[2] + rel[2] - len0 - len1;
        while (bits) {
          const int b0 = __ffsll((long long)bits) - 1;
          bits &= bits - 1;
          int b1 = -1;
          if (bits) {
            b1 = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
          }
          const int pa = b0 + (b0 < len0 ? s0 : (b0 < len0 + len1 ? s1 : s2));
          const int pb = b1 < 0 ? pa : b1 + (b1 < len0 ? s0 : (b1 < len0 + len1 ? s1 : s2));
          const float xa = sx[pa], ya = sy[pa], za = sz[pa], xb = sx[pb], yb = sy[pb], zb = sz[pb];
          const int ia = si[pa], ib = si[pb];
          emit(xa, ya, za, ia);
          if (b1 >= 0) emit(xb, yb, zb, ib);
        }
      } else if (staged) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          for (int p = p0[i] + rel[i]; p < p1[i] + rel[i]; ++p) emit(sx[p], sy[p], sz[p], si[p]);
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          for (int p = p0[i]; p < p1[i]; ++p) {
            const float4 sp = sorted_s[p];
            emit(sp.x, sp.y, sp.z, __float_as_int(sp.w));
          }
      }
    } else if (valid && !skip && slot >= glo && slot < ghi && tot2 != tot && j == NSUB - 1) {
      hits[q_start - gbase + tot] = ~0ull;
      rows[q_start - gbase + tot] = 0xff;
    }
    __syncthreads();
    // ---- one thread per hit: rank inside the segment by counting, then the final slot (or the row buffer)
    const int group_hits = skip ? 0 : offs[ghi] - gbase;
    for (int e = tid; e < group_hits; e += L::THREADS) {
      const int r = rows[e];
      if (r == 0xff) continue;
      const int a = offs[r] - gbase, len = offs[r + 1] - offs[r];  // both even
      const unsigned long long key = hits[e];
      const ulonglong2* seg = reinterpret_cast<const ulonglong2*>(hits + a);
      int rank = 0;
      int jj = 0;
      for (; jj + 4 <= len / 2; jj += 4) {
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
      if (rank < width) {
        if (use_rowbuf) rowbuf[r * width + rank] = (unsigned int)(key & 0xffffffffull);
        else out[(int64_t)orig[r] * width + rank] = (int64_t)(unsigned int)(key & 0xffffffffull);
      }
    }
    if (use_rowbuf) {
      __syncthreads();
      // whole rows leave as contiguous runs: consecutive lanes, consecutive 16-byte pieces of a row
      if ((width & 1) == 0) {
        const int w2 = width >> 1, total_pairs = rows_here * w2;
        const float inv = 1.0f / (float)w2;
        for (int i = tid; i < total_pairs; i += L::THREADS) {
          int r = (int)((float)i * inv);
          r = r * w2 > i ? r - 1 : ((r + 1) * w2 <= i ? r + 1 : r);
          const int cc = (i - r * w2) * 2;
          const int cnt = qtot[r];
          const uint2 v = *reinterpret_cast<const uint2*>(rowbuf + r * width + cc);
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
          out[(int64_t)orig[r] * width + cc] = cc < qtot[r] ? (long long)rowbuf[r * width + cc] : (long long)pad_value;
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

// max / max over the per-block (max hits per query, hits per block) pairs -> hdr
__global__ __launch_bounds__(1024) void reduce_stats_kernel(const int32_t* __restrict__ blk_stats,
                                                            int blocks, RadiusHdr* __restrict__ hdr) {
  __shared__ int sh[2][1024 / WAVE];
  int mx = 0, ms = 0;
  for (int i = threadIdx.x; i < blocks; i += 1024) {
    mx = max(mx, blk_stats[2 * i]);
    ms = max(ms, blk_stats[2 * i + 1]);
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    mx = max(mx, __shfl_xor(mx, d, WAVE));
    ms = max(ms, __shfl_xor(ms, d, WAVE));
  }
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    sh[0][threadIdx.x / WAVE] = mx;
    sh[1][threadIdx.x / WAVE] = ms;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < 1024 / WAVE; ++i) {
      mx = max(mx, sh[0][i]);
      ms = max(ms, sh[1][i]);
    }
    hdr->max_count = (unsigned)mx;
    hdr->max_block_hits = (unsigned)ms;
  }
}

__global__ void pad_fill_kernel(int64_t* __restrict__ out, int64_t n, int64_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

template <int RQ>
int launch_count(const RadiusWs& w, const float4* sorted_q, int64_t nq, int nb, const int32_t* start_s, float r2,
                 bool mono, hipStream_t stream) {
  using L = TravLds<RQ>;
  const int blocks = (int)((nq + RQ - 1) / RQ);
  const int grid = (blocks + 7) / 8 * 8;
  KernelTimer timer("radius_count", stream);
  hipLaunchKernelGGL((traverse_kernel<RQ, false, true>), dim3(grid), dim3(L::THREADS), L::count_bytes(nb <= L::TABLE_MAX ? nb : 0), stream, sorted_q,
                     (int)nq, w.q_off, nb, w.grids, start_s, w.sorted_s, r2, w.q_count, w.q_rng, w.q_mask, w.blk_stats,
                     0, 0, (int64_t)0, (int64_t*)nullptr, 0, (unsigned long long*)nullptr, (unsigned char*)nullptr, mono ? 1 : 0);
  hipLaunchKernelGGL(reduce_stats_kernel, dim3(1), dim3(1024), 0, stream, w.blk_stats, blocks, w.hdr);
  GR_LAUNCH_CHECK();
  return GR_OK;
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

struct FusedCfg {
  int rq;      // queries per block: 64 or 128
  int rowbuf;  // rows leave through an LDS row buffer as contiguous 16-byte pieces (else: one 8-byte store per hit)
  int per_q;   // key slots per query in a block's key area
};
inline FusedCfg fused_cfg() {
  static const FusedCfg cfg = [] {
    FusedCfg c{128, 1, 28};
    if (const char* e = getenv("GR_RADIUS_FUSED_RQ")) c.rq = atoi(e) == 64 ? 64 : 128;
    if (const char* e = getenv("GR_RADIUS_FUSED_ROWBUF")) c.rowbuf = atoi(e) != 0;
    if (const char* e = getenv("GR_RADIUS_FUSED_SLOTS")) c.per_q = max(8, min(512, atoi(e)));
    return c;
  }();
  return cfg;
}

template <int RQ, bool ROWBUF>
int launch_fused_t(const RadiusWs& w, const float4* sorted_q, int64_t nq, int64_t ns, int nb, const int32_t* start_s,
                   float r2, int64_t width, int per_q, int64_t* out, bool mono, hipStream_t stream) {
  using L = FusedLds<RQ>;
  const int blocks = (int)((nq + RQ - 1) / RQ);
  const int grid = (blocks + 7) / 8 * 8;
  const int tcap = nb <= L::TABLE_MAX ? nb : 0;
  // key area: `per_q` slots per query, never less than two full rows, within the 160 KB of a CU
  int cap = max(per_q * RQ, (int)(2 * width + 2));
  cap = (cap + 15) / 16 * 16;
  while (L::total((int)width, cap, tcap) > 160 * 1024 && cap > 64) cap -= 16;
  const size_t region = L::region_bytes((int)width);
  const size_t lds = L::total((int)width, cap, tcap);
  GR_REQUIRE(lds <= 160 * 1024, "radius_search: neighbor_limit %lld does not fit the single-pass kernel", (long long)width);
  auto kern = fused_kernel<RQ, ROWBUF>;
  if (lds > 64 * 1024)
    GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  {
    KernelTimer timer("radius_fused", stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(L::THREADS), lds, stream, sorted_q, (int)nq, w.q_off, nb, w.grids, start_s,
                       w.sorted_s, r2, w.blk_stats, (int)width, ns, out, cap, (int)region, mono ? 1 : 0);
  }
  hipLaunchKernelGGL(reduce_stats_kernel, dim3(1), dim3(1024), 0, stream, w.blk_stats, blocks, w.hdr);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

inline bool fused_fits(int64_t width) {
  // the row buffer / key area of the largest configuration must fit next to the candidate planes
  return width >= 1 && FusedLds<64>::total((int)width, (int)((2 * width + 2 + 15) / 16 * 16), 0) <= 160 * 1024;
}

int launch_fused(const RadiusWs& w, const float4* sorted_q, int64_t nq, int64_t ns, int nb, const int32_t* start_s,
                 float r2, int64_t width, int64_t* out, bool mono, hipStream_t stream) {
  const FusedCfg c = fused_cfg();
  int rq = c.rq;
  if (rq == 128 && FusedLds<128>::total((int)width, (int)((2 * width + 2 + 15) / 16 * 16), 0) > 160 * 1024) rq = 64;
  if (rq == 128)
    return c.rowbuf ? launch_fused_t<128, true>(w, sorted_q, nq, ns, nb, start_s, r2, width, c.per_q, out, mono, stream)
                    : launch_fused_t<128, false>(w, sorted_q, nq, ns, nb, start_s, r2, width, c.per_q, out, mono, stream);
  return c.rowbuf ? launch_fused_t<64, true>(w, sorted_q, nq, ns, nb, start_s, r2, width, c.per_q, out, mono, stream)
                  : launch_fused_t<64, false>(w, sorted_q, nq, ns, nb, start_s, r2, width, c.per_q, out, mono, stream);
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
    if (!reuse) bbox_block_offsets(tmp + batch + 1, tmp + 2 * (batch + 1), (int)batch);
    GR_HIP(hipMemcpyAsync(w.q_off, tmp, sizeof(int32_t) * (reuse ? 1 : 3) * (batch + 1), hipMemcpyHostToDevice, stream));
  }
  const int nb = (int)batch;
  int32_t* cnt_s = w.cnt;
  int32_t* cnt_q = same ? w.cnt : w.cnt + (w.ccap + 1);
  int32_t* start_s = w.start;
  int32_t* start_q = same ? w.start : w.start + (w.ccap + 1);
  int rc = GR_OK;
  KernelTimer bin_timer("radius_bin", stream);  // bbox .. scatter (nothing is launched when the grid is reused in a self-search)
  if (!reuse) {
    // ---- supports (and, in the same launches, the queries): bbox, grid, histogram, scan, scatter
    const int rows = same ? 1 : 2;
    GR_HIP(hipMemsetAsync(w.cnt, 0, sizeof(int32_t) * rows * (w.ccap + 1), stream));
    {
      int rcb = compute_bbox(s, h_offsets + batch + 1, h_offsets + 2 * (batch + 1), w.s_off, nb, w.bbox, w.blk_off, stream, true);
      if (rcb != GR_OK) return rcb;
    }
    // x sub-cells per cell: 2 measured best end to end (count pass 0.166 -> 0.157 ms; 8 gives 0.150 ms but the scan and the
    // scatter over an 8x larger cell table take the difference back)
    constexpr int xk_max = 2;
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(256), 0, stream, w.bbox, w.s_off, nb, radius, xk_max, w.grids, w.hdr);
    const int nq_bin = same ? 0 : (int)nq;
    hipLaunchKernelGGL(bin_count_kernel, dim3((ns + nq_bin + 255) / 256), dim3(256), 0, stream, s, (int)ns, q, nq_bin,
                       w.s_off, w.q_off, nb, w.grids, w.s_cell, w.q_cell, w.s_rank, w.q_rank, cnt_s, cnt_q);
    GR_LAUNCH_CHECK();
    rc = exclusive_scan_i32(w.cnt, w.start, w.ccap + 1, rows, w.ccap + 1, w.scan_ws, nullptr, stream,
                            &w.hdr->total_cells);
    if (rc != GR_OK) return rc;
    hipLaunchKernelGGL(scatter_kernel, dim3((ns + nq_bin + 255) / 256), dim3(256), 0, stream, s, (int)ns, q, nq_bin,
                       w.s_cell, w.q_cell, start_s, start_q, w.s_rank, w.q_rank, w.sorted_s, w.sorted_q);
  } else if (!same) {
    // ---- the support grid is in place: only the queries are binned into it
    GR_HIP(hipMemsetAsync(cnt_q, 0, sizeof(int32_t) * (w.ccap + 1), stream));
    hipLaunchKernelGGL(bin_count_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, s, 0, q, (int)nq, w.s_off,
                       w.q_off, nb, w.grids, w.s_cell, w.q_cell, w.s_rank, w.q_rank, cnt_s, cnt_q);
    GR_LAUNCH_CHECK();
    rc = exclusive_scan_i32(cnt_q, start_q, w.ccap + 1, 1, w.ccap + 1, w.scan_ws, nullptr, stream, &w.hdr->total_cells);
    if (rc != GR_OK) return rc;
    hipLaunchKernelGGL(scatter_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, s, 0, q, (int)nq, w.s_cell,
                       w.q_cell, start_s, start_q, w.s_rank, w.q_rank, w.sorted_s, w.sorted_q);
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
  rc = launch_count<RT>(w, P.sorted_q, nq, P.nb, P.start_s, P.r2, same, stream);
  if (rc != GR_OK) return rc;
  // the read-back lands in pinned memory (a copy into pageable memory is staged and synchronised by the runtime on top of
  // the synchronise below)
  RadiusHdr* h_pinned = static_cast<RadiusHdr*>(pinned_scratch(3, sizeof(RadiusHdr)));
  GR_REQUIRE(h_pinned != nullptr, "pinned read-back buffer could not be allocated");
  GR_HIP(hipMemcpyAsync(h_pinned, w.hdr, sizeof(RadiusHdr), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  const RadiusHdr h = *h_pinned;
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
  return launch_fill<RT>(w, sorted_q, nq, ns, (int)batch, r2, width, width, h_info[1], out, stream);
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
  RadiusHdr* h_pinned = static_cast<RadiusHdr*>(pinned_scratch(3, sizeof(RadiusHdr)));
  GR_REQUIRE(h_pinned != nullptr, "pinned read-back buffer could not be allocated");
  const char* force2 = getenv("GR_RADIUS_TWO_PASS");
  bool fused = fused_fits(limit) && !(force2 && force2[0] == '1');
  if (fused) {
    rc = launch_fused(w, P.sorted_q, nq, ns, P.nb, P.start_s, P.r2, limit, out, P.same, stream);
    if (rc != GR_OK) return rc;
    GR_HIP(hipMemcpyAsync(h_pinned, w.hdr, sizeof(RadiusHdr), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    h_info[0] = h_pinned->max_count;
    h_info[2] = P.same ? 1 : 0;
    h_info[3] = h_pinned->total_cells;
    if (h_pinned->max_block_hits == 0) {  // no query overflowed its block's key area: `out` is complete
      h_info[4] = 1;
      return GR_OK;
    }
  }
  // two passes (a neighbourhood too dense for the key area, or a limit too wide for the row buffer): count, then fill the
  // first min(max_count, limit) columns of the same (nq, limit) rows
  rc = launch_count<RT>(w, P.sorted_q, nq, P.nb, P.start_s, P.r2, P.same, stream);
  if (rc != GR_OK) return rc;
  GR_HIP(hipMemcpyAsync(h_pinned, w.hdr, sizeof(RadiusHdr), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  const RadiusHdr h = *h_pinned;
  h_info[0] = h.max_count;
  h_info[1] = h.max_block_hits;
  h_info[2] = P.same ? 1 : 0;
  h_info[3] = h.total_cells;
  h_info[4] = 0;
  const int64_t width = h.max_count < (uint64_t)limit ? (int64_t)h.max_count : limit;
  if (width == 0) return GR_OK;
  return launch_fill<RT>(w, P.sorted_q, nq, ns, P.nb, P.r2, width, limit, h.max_block_hits, out, stream);
}
