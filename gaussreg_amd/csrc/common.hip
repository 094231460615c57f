// Error plumbing + a small multi-row exclusive scan used by the binning kernels.
#include "common.hpp"

namespace gr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------ scan
constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;  // 2048 elements per block

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    int t = __shfl_up(v, d, WAVE);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, *total
__device__ inline int block_excl_scan(int v, int* total) {
  __shared__ int wsum[SCAN_T / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int inc = wave_incl_scan(v, lane);
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_T / WAVE; ++i) {
    int s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// phase 1: per-tile local exclusive scan, tile sums -> partial[row][tile]
__global__ __launch_bounds__(SCAN_T) void scan_tiles_kernel(const int32_t* __restrict__ in,
                                                            int32_t* __restrict__ out, int64_t n,
                                                            int64_t row_stride,
                                                            int32_t* __restrict__ partial,
                                                            int tiles) {
  const int row = blockIdx.y;
  const int64_t tile0 = (int64_t)blockIdx.x * SCAN_TILE;
  const int32_t* src = in + row * row_stride;
  int32_t* dst = out + row * row_stride;
  int v[SCAN_ITEMS];
  int sum = 0;
  const int64_t base = tile0 + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? src[base + k] : 0;
    sum += v[k];
  }
  int tot;
  int ex = block_excl_scan(sum, &tot);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) dst[base + k] = ex;
    ex += v[k];
  }
  if (threadIdx.x == 0) partial[(int64_t)row * tiles + blockIdx.x] = tot;
}

// phase 2: one block per row scans the tile sums in place (exclusive), writes row total
__global__ __launch_bounds__(SCAN_T) void scan_partials_kernel(int32_t* __restrict__ partial,
                                                               int tiles,
                                                               int32_t* __restrict__ total) {
  int32_t* p = partial + (int64_t)blockIdx.x * tiles;
  int carry = 0;
  for (int t0 = 0; t0 < tiles; t0 += SCAN_T) {
    int i = t0 + threadIdx.x;
    int v = i < tiles ? p[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot);
    if (i < tiles) p[i] = ex + carry;
    carry += tot;
  }
  if (total && threadIdx.x == 0) total[blockIdx.x] = carry;
}

// phase 3: add tile offsets
__global__ __launch_bounds__(SCAN_T) void scan_add_kernel(int32_t* __restrict__ out, int64_t n,
                                                          int64_t row_stride,
                                                          const int32_t* __restrict__ partial,
                                                          int tiles) {
  const int row = blockIdx.y;
  const int add = partial[(int64_t)row * tiles + blockIdx.x];
  if (add == 0) return;
  int32_t* dst = out + row * row_stride;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) dst[base + k] += add;
}

// ------------------------------------------------------------------ per-cloud bounding boxes
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ pts, int n,
                                                   const int32_t* __restrict__ off, int nb,
                                                   uint32_t* __restrict__ bbox) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  int b = -1;
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  if (valid) {
    b = find_batch(off, nb, i);
#pragma unroll
    for (int k = 0; k < 3; ++k) lo[k] = hi[k] = f2ord(pts[3 * (int64_t)i + k]);
  }
  // wave-uniform batch id?  (invalid lanes adopt the first lane's id)
  const int b0 = __shfl(b, 0, WAVE);
  const bool uniform = __all(!valid || b == b0);
  if (uniform) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], d, WAVE));
        hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], d, WAVE));
      }
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && b0 >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        atomicMin(&bbox[b0 * 6 + k], lo[k]);
        atomicMax(&bbox[b0 * 6 + 3 + k], hi[k]);
      }
    }
  } else if (valid) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      atomicMin(&bbox[b * 6 + k], lo[k]);
      atomicMax(&bbox[b * 6 + 3 + k], hi[k]);
    }
  }
}

__global__ void bbox_init_kernel(uint32_t* __restrict__ bbox, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb * 6) bbox[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
}


int compute_bbox(const float* pts, int n, const int32_t* off_dev, int nb, uint32_t* bbox_dev,
                 hipStream_t stream) {
  if (nb <= 0) return GR_OK;
  hipLaunchKernelGGL(bbox_init_kernel, dim3((nb * 6 + 255) / 256), dim3(256), 0, stream, bbox_dev, nb);
  if (n > 0)
    hipLaunchKernelGGL(bbox_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pts, n, off_dev, nb, bbox_dev);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

size_t scan_ws_ints(int64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1; }

int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int rows, int64_t row_stride,
                       int32_t* scan_ws, int32_t* total, hipStream_t stream) {
  if (n <= 0 || rows <= 0) return GR_OK;
  const int tiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles, rows), dim3(SCAN_T), 0, stream, in, out, n,
                     row_stride, scan_ws, tiles);
  hipLaunchKernelGGL(scan_partials_kernel, dim3(rows), dim3(SCAN_T), 0, stream, scan_ws, tiles,
                     total);
  if (tiles > 1)
    hipLaunchKernelGGL(scan_add_kernel, dim3(tiles, rows), dim3(SCAN_T), 0, stream, out, n,
                       row_stride, scan_ws, tiles);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

}  // namespace gr

extern "C" const char* gr_last_error(void) { return gr::g_err; }
extern "C" int gr_version(void) { return 1000; }
