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
//   bbox        per-cloud bounding box of the supports (wave-reduced ordered-uint atomics)
//   grid_setup  per-cloud cell edge (>= radius, coarsened so cells <= max(4096, 4 n_b)), dims, bases
//   bin_count   cell id per support / per query + per-cell histogram
//   scan        exclusive scan of the histograms (common.hip)
//   scatter     counting-sort supports and queries into cell order as float4 {x,y,z,orig index}
//   count       one thread per (cell-ordered) query: hits per query, max over queries  -> host
//   fill        same traversal; hits insertion-sorted by (d, index) in LDS segments sized by the
//               count pass; rows staged in LDS and written as contiguous int64 runs
#include <vector>

#include "common.hpp"

namespace gr {
namespace {

struct BatchGrid {
  double org[3];
  double inv_cell;
  int dim[3];
  int cell_base;
};

struct RadiusHdr {
  unsigned int max_count;
  unsigned int max_block_hits;
  int total_cells;
  int pad;
};

constexpr int RT = 256;  // threads per block in count/fill (block <-> 256 consecutive sorted queries)

struct RadiusWs {
  RadiusHdr* hdr;
  int32_t* q_off;
  int32_t* s_off;
  uint32_t* bbox;
  BatchGrid* grids;
  int32_t* s_cell;
  int32_t* q_cell;
  int32_t* cnt;    // [2][ccap+1]
  int32_t* start;  // [2][ccap+1]
  int32_t* scan_ws;
  float4* sorted_s;
  float4* sorted_q;
  int32_t* q_count;
  int64_t ccap;
  size_t bytes;
};

RadiusWs carve(void* ws, int64_t nq, int64_t ns, int64_t batch) {
  RadiusWs w;
  Carver c(ws);
  w.ccap = 4096 * batch + 4 * ns;
  w.hdr = c.take<RadiusHdr>(1);
  w.q_off = c.take<int32_t>(batch + 1);
  w.s_off = c.take<int32_t>(batch + 1);
  w.bbox = c.take<uint32_t>(batch * 6);
  w.grids = c.take<BatchGrid>(batch);
  w.s_cell = c.take<int32_t>(ns);
  w.q_cell = c.take<int32_t>(nq);
  w.cnt = c.take<int32_t>(2 * (w.ccap + 1));
  w.start = c.take<int32_t>(2 * (w.ccap + 1));
  w.scan_ws = c.take<int32_t>(2 * scan_ws_ints(w.ccap + 1));
  w.sorted_s = c.take<float4>(ns);
  w.sorted_q = c.take<float4>(nq);
  w.q_count = c.take<int32_t>(nq);
  w.bytes = c.used();
  return w;
}

// ---------------------------------------------------------------- grid setup
__global__ void grid_setup_kernel(const uint32_t* __restrict__ bbox,
                                  const int32_t* __restrict__ s_off, int nb, float radius,
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
    g.cell_base = 0;
    grids[b] = g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int base = 0;
    for (int b = 0; b < nb; ++b) {
      grids[b].cell_base = base;
      base += grids[b].dim[0] * grids[b].dim[1] * grids[b].dim[2];
    }
    hdr->total_cells = base;
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
    double u = cell_coord(p[k], g.org[k], g.inv_cell);
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
    int32_t* __restrict__ cnt_s, int32_t* __restrict__ cnt_q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) {
    const int b = find_batch(s_off, nb, i);
    const int c = clamped_cell(grids[b], s[3 * (int64_t)i], s[3 * (int64_t)i + 1], s[3 * (int64_t)i + 2]);
    s_cell[i] = c;
    atomicAdd(&cnt_s[c], 1);
  } else if (i < ns + nq) {
    const int j = i - ns;
    const int b = find_batch(q_off, nb, j);
    const int c = clamped_cell(grids[b], q[3 * (int64_t)j], q[3 * (int64_t)j + 1], q[3 * (int64_t)j + 2]);
    q_cell[j] = c;
    atomicAdd(&cnt_q[c], 1);
  }
}

__global__ __launch_bounds__(256) void scatter_kernel(
    const float* __restrict__ s, int ns, const float* __restrict__ q, int nq,
    const int32_t* __restrict__ s_cell, const int32_t* __restrict__ q_cell,
    const int32_t* __restrict__ start_s, const int32_t* __restrict__ start_q,
    int32_t* __restrict__ cnt_s, int32_t* __restrict__ cnt_q, float4* __restrict__ sorted_s,
    float4* __restrict__ sorted_q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) {
    const int c = s_cell[i];
    const int slot = start_s[c] + atomicSub(&cnt_s[c], 1) - 1;
    sorted_s[slot] = make_float4(s[3 * (int64_t)i], s[3 * (int64_t)i + 1], s[3 * (int64_t)i + 2],
                                 __int_as_float(i));
  } else if (i < ns + nq) {
    const int j = i - ns;
    const int c = q_cell[j];
    const int slot = start_q[c] + atomicSub(&cnt_q[c], 1) - 1;
    sorted_q[slot] = make_float4(q[3 * (int64_t)j], q[3 * (int64_t)j + 1], q[3 * (int64_t)j + 2],
                                 __int_as_float(j));
  }
}

// ---------------------------------------------------------------- candidate traversal
// Calls f(dist, support_orig_index) for every support of the query's 27-cell neighbourhood
// with d < r2.
template <typename F>
__device__ inline void for_each_hit(const float4 qp, const BatchGrid& g,
                                    const int32_t* __restrict__ start_s,
                                    const float4* __restrict__ sorted_s, float r2, F&& f) {
  int lo[3], hi[3];
  const float p[3] = {qp.x, qp.y, qp.z};
  bool empty = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double u = cell_coord(p[k], g.org[k], g.inv_cell);
    const double top = (double)(g.dim[k] - 1);
    if (!(u + 1.0 >= 0.0) || !(u - 1.0 <= top)) empty = true;  // also catches NaN
    lo[k] = (int)fmin(fmax(u - 1.0, 0.0), top);
    hi[k] = (int)fmin(fmax(u + 1.0, 0.0), top);
  }
  if (empty) return;
  for (int cz = lo[2]; cz <= hi[2]; ++cz)
    for (int cy = lo[1]; cy <= hi[1]; ++cy) {
      const int base = g.cell_base + g.dim[0] * (cy + g.dim[1] * cz);
      const int p0 = start_s[base + lo[0]];
      const int p1 = start_s[base + hi[0] + 1];
      for (int t = p0; t < p1; ++t) {
        const float4 sp = sorted_s[t];
        // nanoflann.hpp:432-440: result += diff*diff for x, y, z starting from 0
        const float dx = qp.x - sp.x;
        const float dy = qp.y - sp.y;
        const float dz = qp.z - sp.z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < r2) f(d, __float_as_int(sp.w));
      }
    }
}

__global__ __launch_bounds__(RT) void count_kernel(const float4* __restrict__ sorted_q, int nq,
                                                   const int32_t* __restrict__ q_off, int nb,
                                                   const BatchGrid* __restrict__ grids,
                                                   const int32_t* __restrict__ start_s,
                                                   const float4* __restrict__ sorted_s, float r2,
                                                   int32_t* __restrict__ q_count,
                                                   RadiusHdr* __restrict__ hdr) {
  __shared__ int wsum[RT / WAVE];
  const int t = blockIdx.x * RT + threadIdx.x;
  int n = 0;
  if (t < nq) {
    const float4 qp = sorted_q[t];
    const int b = find_batch(q_off, nb, __float_as_int(qp.w));
    const BatchGrid g = grids[b];
    for_each_hit(qp, g, start_s, sorted_s, r2, [&](float, int) { ++n; });
    q_count[t] = n;
  }
  int mx = n, sm = n;
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    mx = max(mx, __shfl_xor(mx, d, WAVE));
    sm += __shfl_xor(sm, d, WAVE);
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  if (lane == 0) {
    wsum[w] = sm;
    if (mx > 0) atomicMax(&hdr->max_count, (unsigned)mx);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
#pragma unroll
    for (int i = 0; i < RT / WAVE; ++i) tot += wsum[i];
    if (tot > 0) atomicMax(&hdr->max_block_hits, (unsigned)tot);
  }
}

// LDS layout of the fill kernel (all dynamic, base 16-B aligned):
//   [offs: RT+1 ints][orig: RT ints][wsum: RT/64 ints][pad to LDS_FIXED][segments: u64 ...]
constexpr size_t LDS_FIXED = ((4 * (RT + 1) + 4 * RT + 4 * (RT / WAVE)) + 15) / 16 * 16;

__global__ void pad_fill_kernel(int64_t* __restrict__ out, int64_t n, int64_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

template <bool SEG_IN_LDS>
__global__ __launch_bounds__(RT) void fill_kernel(
    const float4* __restrict__ sorted_q, int nq, const int32_t* __restrict__ q_off, int nb,
    const BatchGrid* __restrict__ grids, const int32_t* __restrict__ start_s,
    const float4* __restrict__ sorted_s, float r2, const int32_t* __restrict__ q_count, int width,
    int64_t pad_value, int64_t* __restrict__ out, unsigned long long* __restrict__ gseg,
    int64_t gseg_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* offs = reinterpret_cast<int*>(smem);
  int* orig = offs + (RT + 1);
  int* wsum = orig + RT;
  unsigned long long* seg_lds = reinterpret_cast<unsigned long long*>(smem + LDS_FIXED);

  const int t = blockIdx.x * RT + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const bool valid = t < nq;
  const int cnt = valid ? q_count[t] : 0;
  // block exclusive scan of cnt
  int inc = cnt;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    int v = __shfl_up(inc, d, WAVE);
    if (lane >= d) inc += v;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < RT / WAVE; ++i)
    if (i < w) base += wsum[i];
  const int my_off = base + inc - cnt;
  offs[threadIdx.x] = my_off;
  if (threadIdx.x == RT - 1) offs[RT] = my_off + cnt;

  float4 qp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) qp = sorted_q[t];
  orig[threadIdx.x] = valid ? __float_as_int(qp.w) : -1;

  unsigned long long* seg =
      SEG_IN_LDS ? (seg_lds + my_off) : (gseg + (int64_t)blockIdx.x * gseg_stride + my_off);
  if (valid && cnt > 0) {
    const int b = find_batch(q_off, nb, __float_as_int(qp.w));
    const BatchGrid g = grids[b];
    int n = 0;
    for_each_hit(qp, g, start_s, sorted_s, r2, [&](float d, int idx) {
      // key orders by (distance, index); d >= 0 so its bit pattern is monotone
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)idx;
      int j = n;
      while (j > 0) {
        const unsigned long long prev = seg[j - 1];
        if (prev <= key) break;
        seg[j] = prev;
        --j;
      }
      seg[j] = key;
      ++n;
    });
  }
  __syncthreads();
  // cooperative write-out: rows of this block, `width` int64 each, contiguous per row
  const int rows = min(RT, nq - blockIdx.x * RT);
  const unsigned long long* segb =
      SEG_IN_LDS ? seg_lds : (gseg + (int64_t)blockIdx.x * gseg_stride);
  const int total = rows * width;
  for (int e = threadIdx.x; e < total; e += RT) {
    const int r = e / width;
    const int c = e - r * width;
    const int o = offs[r];
    const int n = offs[r + 1] - o;
    const int64_t v = c < n ? (int64_t)(unsigned int)(segb[o + c] & 0xffffffffull) : pad_value;
    out[(int64_t)orig[r] * width + c] = v;
  }
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
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_info != nullptr, "h_info is null");
  h_info[0] = h_info[1] = h_info[2] = h_info[3] = 0;
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
  // offsets (host -> device)
  {
    std::vector<int32_t> tmpv(2 * (batch + 1));
    int32_t* tmp = tmpv.data();
    tmp[0] = 0;
    tmp[batch + 1] = 0;
    for (int64_t b = 0; b < batch; ++b) {
      tmp[b + 1] = tmp[b] + (int32_t)h_q_lengths[b];
      tmp[batch + 1 + b + 1] = tmp[batch + 1 + b] + (int32_t)h_s_lengths[b];
    }
    GR_HIP(hipMemcpyAsync(w.q_off, tmp, sizeof(int32_t) * (batch + 1), hipMemcpyHostToDevice, stream));
    GR_HIP(hipMemcpyAsync(w.s_off, tmp + batch + 1, sizeof(int32_t) * (batch + 1), hipMemcpyHostToDevice, stream));
  }
  const int nb = (int)batch;
  const int rows = same ? 1 : 2;
  GR_HIP(hipMemsetAsync(w.cnt, 0, sizeof(int32_t) * rows * (w.ccap + 1), stream));
  {
    int rcb = compute_bbox(s, (int)ns, w.s_off, nb, w.bbox, stream);
    if (rcb != GR_OK) return rcb;
  }
  hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(256), 0, stream, w.bbox, w.s_off, nb, radius, w.grids, w.hdr);
  int32_t* cnt_s = w.cnt;
  int32_t* cnt_q = same ? w.cnt : w.cnt + (w.ccap + 1);
  int32_t* start_s = w.start;
  int32_t* start_q = same ? w.start : w.start + (w.ccap + 1);
  const int nq_bin = same ? 0 : (int)nq;
  hipLaunchKernelGGL(bin_count_kernel, dim3((ns + nq_bin + 255) / 256), dim3(256), 0, stream, s, (int)ns, q, nq_bin,
                     w.s_off, w.q_off, nb, w.grids, w.s_cell, w.q_cell, cnt_s, cnt_q);
  GR_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(w.cnt, w.start, w.ccap + 1, rows, w.ccap + 1, w.scan_ws, nullptr, stream);
  if (rc != GR_OK) return rc;
  hipLaunchKernelGGL(scatter_kernel, dim3((ns + nq_bin + 255) / 256), dim3(256), 0, stream, s, (int)ns, q, nq_bin,
                     w.s_cell, w.q_cell, start_s, start_q, cnt_s, cnt_q, w.sorted_s, w.sorted_q);
  const float4* sorted_q = same ? w.sorted_s : w.sorted_q;
  const float r2 = radius * radius;  // radius_neighbors_cpu.cpp:12 (fp32 product)
  hipLaunchKernelGGL(count_kernel, dim3((nq + RT - 1) / RT), dim3(RT), 0, stream, sorted_q, (int)nq, w.q_off, nb,
                     w.grids, start_s, w.sorted_s, r2, w.q_count, w.hdr);
  GR_LAUNCH_CHECK();
  RadiusHdr h;
  GR_HIP(hipMemcpyAsync(&h, w.hdr, sizeof(h), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
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
  const int64_t max_block_hits = h_info[1];
  const size_t lds_fixed = LDS_FIXED;
  const size_t lds_seg = (size_t)max_block_hits * 8;
  const int blocks = (int)((nq + RT - 1) / RT);
  if (lds_fixed + lds_seg <= 64 * 1024) {
    hipLaunchKernelGGL(fill_kernel<true>, dim3(blocks), dim3(RT), lds_fixed + lds_seg, stream, sorted_q, (int)nq,
                       w.q_off, (int)batch, w.grids, w.start, w.sorted_s, r2, w.q_count, (int)width, ns, out,
                       (unsigned long long*)nullptr, (int64_t)0);
  } else if (lds_fixed + lds_seg <= 160 * 1024) {
    GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(fill_kernel<true>, dim3(blocks), dim3(RT), lds_fixed + lds_seg, stream, sorted_q, (int)nq,
                       w.q_off, (int)batch, w.grids, w.start, w.sorted_s, r2, w.q_count, (int)width, ns, out,
                       (unsigned long long*)nullptr, (int64_t)0);
  } else {
    // very dense neighbourhoods: segments live in a scratch allocation owned by this call
    unsigned long long* gseg = nullptr;
    GR_HIP(hipMallocAsync(reinterpret_cast<void**>(&gseg), (size_t)blocks * (size_t)max_block_hits * 8, stream));
    hipLaunchKernelGGL(fill_kernel<false>, dim3(blocks), dim3(RT), lds_fixed, stream, sorted_q, (int)nq, w.q_off,
                       (int)batch, w.grids, w.start, w.sorted_s, r2, w.q_count, (int)width, ns, out, gseg,
                       max_block_hits);
    GR_HIP(hipFreeAsync(gseg, stream));
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}
