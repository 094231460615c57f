// Voxel-grid barycentre subsampling on MI355X, bit-identical to the reference.
//
// Replaces  geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75
// (std::unordered_map voxel hashing, single thread) with:
//   bbox      per-cloud min/max corner                      (cloud.cpp:4-37)
//   [host]    origin / NX / NY per cloud, same fp32 expressions as :11-20 (B clouds -> trivial)
//   keys      key = iX + NX*iY + NX*NY*iZ per point          (:32-35), composite (cloud, key)
//   sort      stable radix sort of (key, point index)  -> points of one voxel are contiguous and
//             in INPUT ORDER, which is what makes the fp32 sums reproducible
//   cells     one thread per voxel run: sequential fp32 accumulate in input order
//             (grid_subsampling_cpu.h:17-20), barycentre = sum * float(1.0 / count)  (:46)
//   order     GR_ORDER_CELL: emit in (cloud, key) order.
//             GR_ORDER_REFERENCE: rank voxels by first occurrence, replay those keys through a
//             real std::unordered_map<size_t,int> on the host (the container whose iteration
//             order IS the reference's output order, :44-47) and gather by that permutation.
// Compiled with -ffp-contract=off.
#include <cmath>
#include <vector>

#include <algorithm>
#include <atomic>

#include "common.hpp"

namespace gr {
namespace {

struct CloudGrid {
  float ox, oy, oz, v;
  unsigned long long nx, nxny;
  int sh, sh_fo;  // bucket sorts: (cells - 1) >> sh < 512 (voxel keys), (points - 1) >> sh_fo < 512 (first occurrences)
};

struct GridWs {
  int32_t* off;         // [B+1]
  uint32_t* bbox;       // [B*6]
  int32_t* blk_off;     // [B+1]
  int32_t* ticket;      // [1] zeroed with the first upload: the last bbox workgroup posts the boxes to the host
  CloudGrid* grids;     // [B]
  int32_t* cblk;        // [B+1] first 256-position workgroup of every cloud (workgroups do not straddle clouds); uploaded with grids
  uint64_t* keys_a;     // [N]
  uint64_t* keys_b;     // [N]
  int32_t* vals_a;      // [N]
  int32_t* vals_b;      // [N]
  int32_t* flags;       // [N]  head flags, then reused
  int32_t* scan;        // [N]
  int32_t* scan_ws;     // scan scratch
  int32_t* totals;      // [2]
  float* bary;          // [N*3] (cell order)
  int32_t* first_idx;   // [N]
  uint64_t* cell_key;   // [N]  raw voxel key per cell
  int32_t* cell_batch;  // [N]
  int32_t* m_b;         // [B + 1]
  int32_t* cell_of_rank;  // [N]
  uint64_t* keys_fo;    // [N]  keys in first-occurrence order
  int32_t* perm;        // [N]
  void* sort_temp;
  size_t sort_temp_bytes;
  void* ho_ws;          // device evaluation of the unordered_map iteration order (hash_order_device.hip)
  size_t ho_bytes;
  void* ds_table;       // bucket sort (depth_sort.hip): histograms / offsets, sized for rows <= ds_rows (cloud, chunk) pairs
  size_t ds_bytes;
  int64_t ds_rows;
  uint32_t* ds_range;   // [2 B] {smallest field, shift} per cloud
  uint32_t* ds_range_fo;  // [2 B] the same for the first-occurrence sort
  int32_t* cell_off;    // [B + 1] first cell (run) of every cloud, on the device (cloud_counts_kernel)
  int32_t* ds_nvalid;   // [B]
  int32_t* ds_ovf;      // [1] a bucket did not fit
  size_t bytes;
};

// rows of the bucket sort's tables the workspace holds: (cloud, 2048-point chunk of the LONGEST cloud) pairs -- up to twice
// what the points need, so a batch of clouds of similar length fits and a very ragged one takes the general sort
inline int64_t ds_rows_cap(int64_t n, int64_t batch) { return 2 * ((n + 2047) / 2048) + 2 * batch; }

GridWs carve(void* ws, int64_t n, int64_t batch) {
  GridWs w;
  Carver c(ws);
  w.off = c.take<int32_t>(batch + 1);
  w.bbox = c.take<uint32_t>(batch * 6);
  w.blk_off = c.take<int32_t>(batch + 1);
  w.ticket = c.take<int32_t>(1);
  w.grids = c.take<CloudGrid>(batch);
  w.cblk = c.take<int32_t>(batch + 1);
  w.keys_a = c.take<uint64_t>(n);
  w.keys_b = c.take<uint64_t>(n);
  w.vals_a = c.take<int32_t>(n);
  w.vals_b = c.take<int32_t>(n);
  w.flags = c.take<int32_t>(n);
  w.scan = c.take<int32_t>(n);
  w.scan_ws = c.take<int32_t>(scan_ws_ints(n));
  w.totals = c.take<int32_t>(2);
  w.bary = c.take<float>(3 * n);
  w.first_idx = c.take<int32_t>(n);
  w.cell_key = c.take<uint64_t>(n);
  w.cell_batch = c.take<int32_t>(n);
  w.m_b = c.take<int32_t>(batch + 2);  // [batch] = the total, [batch + 1] = bucket-overflow flag (one read-back for all)
  w.cell_of_rank = c.take<int32_t>(n);
  w.keys_fo = c.take<uint64_t>(n);
  w.perm = c.take<int32_t>(n);
  w.sort_temp_bytes = sort_pairs_temp_bytes(n);
  w.sort_temp = c.take<char>(w.sort_temp_bytes);
  w.ho_bytes = hash_order_device_bytes(n, batch);
  w.ho_ws = c.take<char>(w.ho_bytes);
  w.ds_rows = ds_rows_cap(n, batch);
  w.ds_bytes = align_up((size_t)w.ds_rows * 512 * sizeof(uint16_t), 256) + align_up((size_t)w.ds_rows * 512 * sizeof(uint32_t), 256) +
               align_up((size_t)batch * 512 * sizeof(int32_t), 256) + align_up((size_t)batch * 2 * sizeof(uint32_t), 256) +
               align_up(((size_t)batch * 512 + 1) * sizeof(int32_t), 256) + 512;  // (depth_sort_table_bytes with rows for nchunk * batch)
  w.ds_table = c.take<char>(w.ds_bytes);
  w.ds_range = c.take<uint32_t>(2 * batch);
  w.ds_range_fo = c.take<uint32_t>(2 * batch);
  w.cell_off = c.take<int32_t>(batch + 1);
  w.ds_nvalid = c.take<int32_t>(batch);
  w.ds_ovf = c.take<int32_t>(1);
  w.bytes = c.used();
  return w;
}

// (size_t)floor(x) as g++ emits it on x86-64 for in-range and slightly negative values:
// signed truncation, then reinterpret (grid_subsampling_cpu.cpp:32-34 static_cast<size_t>)
__host__ __device__ inline unsigned long long to_size_t(double f) {
  return (unsigned long long)(long long)f;
}

__global__ __launch_bounds__(256) void keys_kernel(const float* __restrict__ pts, int n,
                                                   const int32_t* __restrict__ off, int nb,
                                                   const CloudGrid* __restrict__ grids,
                                                   int key_bits, uint64_t* __restrict__ keys,
                                                   int32_t* __restrict__ vals, int32_t* __restrict__ fo_flags) {
  // the cloud offsets of small batches are searched in LDS (six dependent global reads per point otherwise)
  __shared__ int32_t s_off[256];
  const bool in_lds = nb + 1 <= 256;
  if (in_lds) {
    if ((int)threadIdx.x <= nb) s_off[threadIdx.x] = off[threadIdx.x];
    __syncthreads();
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (fo_flags) fo_flags[i] = 0;  // first-occurrence flags (reference order), set by cells_kernel
  const int b = in_lds ? find_batch(s_off, nb, i) : find_batch(off, nb, i);
  const CloudGrid g = grids[b];
  const float x = pts[3 * (int64_t)i], y = pts[3 * (int64_t)i + 1], z = pts[3 * (int64_t)i + 2];
  // grid_subsampling_cpu.cpp:32-35: fp32 subtract, fp32 divide, floor
  const unsigned long long ix = to_size_t(floor((double)((x - g.ox) / g.v)));
  const unsigned long long iy = to_size_t(floor((double)((y - g.oy) / g.v)));
  const unsigned long long iz = to_size_t(floor((double)((z - g.oz) / g.v)));
  unsigned long long key = ix + g.nx * iy + g.nxny * iz;
  if (key_bits < 64) key |= (unsigned long long)b << key_bits;  // composite (cloud, key)
  keys[i] = key;
  vals[i] = i;
}

// Bucket-sort front (a batch whose keys fit 26 bits): field = voxel key + 1 (never 0 = "culled" for depth_sort.hip), payload =
// the voxel key; the first workgroup leaves every cloud's {smallest field, shift} and clears the overflow flag.
__global__ __launch_bounds__(256) void keys32_kernel(const float* __restrict__ pts, int n, const int32_t* __restrict__ off, int nb,
                                                     const CloudGrid* __restrict__ grids, uint32_t* __restrict__ field,
                                                     uint32_t* __restrict__ payload, int32_t* __restrict__ fo_flags,
                                                     uint32_t* __restrict__ range, int32_t* __restrict__ ovf,
                                                     uint32_t* __restrict__ range_fo) {
  __shared__ int32_t s_off[256];
  const bool in_lds = nb + 1 <= 256;
  if (in_lds) {
    if ((int)threadIdx.x <= nb) s_off[threadIdx.x] = off[threadIdx.x];
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (int b = threadIdx.x; b < nb; b += 256) {
      range[2 * b] = 1u, range[2 * b + 1] = (uint32_t)grids[b].sh;
      range_fo[2 * b] = 1u, range_fo[2 * b + 1] = (uint32_t)grids[b].sh_fo;
    }
    if (threadIdx.x == 0) *ovf = 0;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (fo_flags) fo_flags[i] = 0;
  const int b = in_lds ? find_batch(s_off, nb, i) : find_batch(off, nb, i);
  const CloudGrid g = grids[b];
  const float x = pts[3 * (int64_t)i], y = pts[3 * (int64_t)i + 1], z = pts[3 * (int64_t)i + 2];
  const unsigned long long ix = to_size_t(floor((double)((x - g.ox) / g.v)));
  const unsigned long long iy = to_size_t(floor((double)((y - g.oy) / g.v)));
  const unsigned long long iz = to_size_t(floor((double)((z - g.oz) / g.v)));
  const uint32_t key = (uint32_t)(ix + g.nx * iy + g.nxny * iz);  // < 2^26 (checked on the host: no negative cell, no wrap)
  field[i] = key + 1u;
  payload[i] = key;
}

// keys (without the cloud id) for points listed in `vals` order
__global__ __launch_bounds__(256) void regather_keys_kernel(const float* __restrict__ pts,
                                                            const int32_t* __restrict__ vals, int n,
                                                            const int32_t* __restrict__ off, int nb,
                                                            const CloudGrid* __restrict__ grids,
                                                            uint64_t* __restrict__ keys) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int64_t i = vals[t];
  const CloudGrid g = grids[find_batch(off, nb, (int)i)];
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  const unsigned long long ix = to_size_t(floor((double)((x - g.ox) / g.v)));
  const unsigned long long iy = to_size_t(floor((double)((y - g.oy) / g.v)));
  const unsigned long long iz = to_size_t(floor((double)((z - g.oz) / g.v)));
  keys[t] = ix + g.nx * iy + g.nxny * iz;
}

__global__ __launch_bounds__(256) void batch_keys_kernel(const int32_t* __restrict__ vals, int n,
                                                         const int32_t* __restrict__ off, int nb,
                                                         uint64_t* __restrict__ bkeys) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) bkeys[t] = (uint64_t)find_batch(off, nb, vals[t]);
}

// does a new (cloud, voxel) run start at sorted position t?
__device__ __forceinline__ int is_head(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                       const int32_t* __restrict__ off, int nb, int composite, int t) {
  if (t == 0) return 1;
  int h = keys[t] != keys[t - 1];
  if (!h && !composite) h = find_batch(off, nb, vals[t]) != find_batch(off, nb, vals[t - 1]);
  return h;
}

// run heads per workgroup of 256 sorted positions.  The cell index of a run = (heads in the workgroups in front: a scan over
// n / 256 values) + (heads in front of it inside its own workgroup: cells_kernel counts those itself) -- the head flags and
// their scan over all n positions (three launches, 100 MB written and read back at 64 x 200 k) are never materialised.
// (Workgroups do not straddle clouds -- cblk[b] = first workgroup of cloud b -- so a cloud's run count is a difference of two
// scan values and nobody has to count inside a workgroup for it.)
// (one thread searches -- seven dependent loads -- and hands the answer to the others through LDS; contains a barrier)
__device__ __forceinline__ int wg_first_position(const int32_t* __restrict__ cblk, const int32_t* __restrict__ off, int nb, int wg,
                                                 int& end, int* s_pair) {
  if (threadIdx.x == 0) {
    const int b = find_batch(cblk, nb, wg);
    s_pair[0] = off[b] + (wg - cblk[b]) * 256;
    s_pair[1] = off[b + 1];
  }
  __syncthreads();
  end = s_pair[1];
  return s_pair[0];
}

__global__ __launch_bounds__(256) void head_count_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, int n,
                                                         const int32_t* __restrict__ off, int nb, int composite,
                                                         const int32_t* __restrict__ cblk, int32_t* __restrict__ blk_cnt) {
  __shared__ int s_w[256 / WAVE];
  __shared__ int s_pair[2];
  int end;
  const int t = wg_first_position(cblk, off, nb, (int)blockIdx.x, end, s_pair) + (int)threadIdx.x;
  const int h = t < end ? is_head(keys, vals, off, nb, composite, t) : 0;
  const int c = __popcll(__ballot(h != 0));
  if ((threadIdx.x & (WAVE - 1)) == 0) s_w[threadIdx.x / WAVE] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// Barycentres: sequential fp32 sums in input order (stable sort => ascending index), one thread per run head -- but the
// gather of the points is done by ALL threads first (one coalesced read of the sorted indices, one gather of the
// coordinates into LDS, every lane busy, two memory round trips per workgroup); the head threads then walk their runs in
// LDS.  (Head threads fetching their own points one after the other through index -> point chains: 0.38 ms of the 2.5 ms of
// a 64 x 200 k call.)  A run that continues past the workgroup's 256 positions is finished from global memory.
__global__ __launch_bounds__(256) void cells_kernel(
    const float* __restrict__ pts, const uint64_t* __restrict__ keys,
    const int32_t* __restrict__ vals, const int32_t* __restrict__ blk_base, int composite,
    const int32_t* __restrict__ cblk, int n_wg, int n, const int32_t* __restrict__ off, int nb,
    int key_bits, float* __restrict__ bary, int32_t* __restrict__ first_idx,
    uint64_t* __restrict__ cell_key, int32_t* __restrict__ cell_batch,
    int32_t* __restrict__ fo_flags, int fo_local) {
  // fo_local: first_idx[cell] = the first point's index INSIDE its cloud, + 1 -- the field of the first-occurrence sort
  // (depth_sort.hip; no flags over the points then)
  __shared__ float s_x[256], s_y[256], s_z[256];
  __shared__ int s_head[256];
  __shared__ int s_wc[256 / WAVE];
  // XCD-aware block order (block b runs on XCD b % 8): every XCD gets one contiguous eighth of the sorted positions, i.e. its
  // own clouds.  The gather below reads a cloud's points at random: in launch order all eight L2s fetched every cloud
  // (measured: 1.2 GB of fetches for 154 MB of points at 64 x 200 k); now each cloud is fetched by one L2.
  const int per_xcd = gridDim.x / 8;  // the grid is padded to a multiple of 8 blocks
  const int blk = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;
  if (blk >= n_wg) return;  // (block-uniform)
  __shared__ int s_pair[2];
  int end;
  const int t = wg_first_position(cblk, off, nb, blk, end, s_pair) + (int)threadIdx.x;
  const int tc = min(t, end - 1);
  const int64_t mine = vals[tc];
  const int h = t < end ? is_head(keys, vals, off, nb, composite, tc) : 1;  // positions past the cloud's end close its last run
  // heads in front of this position inside the workgroup (blk_base: heads in the workgroups in front, see head_count_kernel)
  const unsigned long long hb = __ballot(h != 0 && t < end);
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if (lane == 0) s_wc[wv] = __popcll(hb);
  s_x[threadIdx.x] = pts[3 * mine];
  s_y[threadIdx.x] = pts[3 * mine + 1];
  s_z[threadIdx.x] = pts[3 * mine + 2];
  s_head[threadIdx.x] = h;
  __syncthreads();
  if (t >= end || !h) return;
  int cell = blk_base[blk] + __popcll(hb & ((1ull << lane) - 1ull));
  for (int w2 = 0; w2 < wv; ++w2) cell += s_wc[w2];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  int count = 0;
  int l = threadIdx.x;
  do {
    sx += s_x[l];
    sy += s_y[l];
    sz += s_z[l];
    ++count;
    ++l;
  } while (l < 256 && !s_head[l]);
  if (l == 256) {  // the run may go on in the next workgroup's positions
    int u = t + count;
    while (u < n && !is_head(keys, vals, off, nb, composite, u)) {
      const int64_t i = vals[u];
      sx += pts[3 * i];
      sy += pts[3 * i + 1];
      sz += pts[3 * i + 2];
      ++count;
      ++u;
    }
  }
  // grid_subsampling_cpu.cpp:46: point * (1.0 / count) -- double reciprocal narrowed to float
  const float wgt = (float)(1.0 / (double)count);
  bary[3 * (int64_t)cell] = sx * wgt;
  bary[3 * (int64_t)cell + 1] = sy * wgt;
  bary[3 * (int64_t)cell + 2] = sz * wgt;
  const int first = (int)mine;
  const unsigned long long k = keys[t];
  // composite keys carry the cloud in their high bits: no search through the offsets (six dependent loads per head thread)
  const int b = key_bits < 64 ? (int)(k >> key_bits) : find_batch(off, nb, first);
  first_idx[cell] = fo_local ? first - off[b] + 1 : first;
  cell_batch[cell] = b;
  cell_key[cell] = key_bits < 64 ? (k & ((1ull << key_bits) - 1ull)) : k;
  if (fo_flags) fo_flags[first] = 1;  // (reference order only)
}

// m_b = number of voxel runs of cloud b = runs in front of its successor's first workgroup - runs in front of its own.
// mail (single-workgroup launches only): the nb + 1 words also go to the host's mailbox page, stamped
__global__ void cloud_counts_kernel(const int32_t* __restrict__ cblk, int n_wg, const int32_t* __restrict__ blk_base,
                                    const int32_t* __restrict__ total, int n,
                                    const int32_t* __restrict__ off, int nb,
                                    int32_t* __restrict__ m_b, int32_t* mail, int stamp,
                                    const int32_t* __restrict__ ovf, int32_t* __restrict__ cell_off) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) {
    m_b[nb] = total[0];
    m_b[nb + 1] = ovf ? *ovf : 0;
    if (mail) mail[nb] = total[0], mail[nb + 1] = m_b[nb + 1];
  }
  if (b < nb) {
    const int wa = cblk[b], we = cblk[b + 1];  // the cloud's workgroups
    const int ca = wa < n_wg ? blk_base[wa] : total[0], ce = we < n_wg ? blk_base[we] : total[0];
    m_b[b] = ce - ca;
    cell_off[b] = ca;
    if (b == nb - 1) cell_off[nb] = ce;
    if (mail) mail[b] = ce - ca;
  }
  if (mail) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) mail_post(mail + nb + 2, stamp);
  }
}

// rank cells by first occurrence: rank = (#cells whose first point index is smaller)
__global__ __launch_bounds__(256) void fo_rank_kernel(const int32_t* __restrict__ first_idx,
                                                      const uint64_t* __restrict__ cell_key,
                                                      const int32_t* __restrict__ fo_scan, int m,
                                                      int32_t* __restrict__ cell_of_rank,
                                                      uint64_t* __restrict__ keys_fo) {
  // (XCD-aware block order as in cells_kernel: the reads and writes below are random inside one cloud)
  const int per_xcd = gridDim.x / 8;
  const int c = (((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8) * 256 + threadIdx.x;
  if (c >= m) return;
  const int r = fo_scan[first_idx[c]];
  cell_of_rank[r] = c;
  keys_fo[r] = cell_key[c];
}

// Test switch (gr_grid_subsample_debug_bucket_sort): 0 = every call takes the general radix sort
std::atomic<int> g_grid_bucket_sort{1};
std::atomic<int> g_grid_bucket_fallbacks{0};  // calls that started over because a bucket overflowed (test hook)

inline int bits_for(unsigned long long v) {  // bits needed to represent values in [0, v)
  int b = 0;
  while (b < 64 && (1ull << b) < v) ++b;
  return b;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_grid_subsample_workspace_bytes(int64_t n, int64_t batch) {
  if (n < 0 || batch < 0) return 0;
  return carve(nullptr, n, batch).bytes;
}

static int grid_subsample_impl(const float* points, const int64_t* h_lengths, int64_t n,
                               int64_t batch, float voxel, int order_mode, float* out_points,
                               int64_t* h_out_lengths, int64_t* h_total_m, void* ws,
                               size_t ws_bytes, void* stream_, bool allow_bucket_sort) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_total_m && (batch == 0 || (h_lengths && h_out_lengths)), "null host pointer");
  *h_total_m = 0;
  GR_REQUIRE(n >= 0 && batch >= 0 && n < (1ll << 31) - 1 && batch < (1 << 20), "bad sizes");
  GR_REQUIRE(order_mode == GR_ORDER_REFERENCE || order_mode == GR_ORDER_CELL, "bad order_mode %d", order_mode);
  GR_REQUIRE(voxel > 0.0f && std::isfinite(voxel), "voxel_size must be positive and finite");
  int64_t sum = 0;
  for (int64_t b = 0; b < batch; ++b) {
    GR_REQUIRE(h_lengths[b] >= 0, "negative length");
    sum += h_lengths[b];
    h_out_lengths[b] = 0;
  }
  GR_REQUIRE(sum == n, "lengths sum to %lld, expected %lld", (long long)sum, (long long)n);
  if (n == 0) return GR_OK;
  GridWs w = carve(ws, n, batch);
  if (!ws || ws_bytes < w.bytes) {
    set_error("grid_subsample workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return GR_ERR_WORKSPACE;
  }
  const int nb = (int)batch;
  // Host staging is pinned (pageable copies block the host and leave the device idle meanwhile).  Its first part mirrors
  // the first three arrays of the workspace -- point offsets, bounding boxes (their initial values), bbox workgroup
  // offsets -- so ONE upload replaces two copies and the bbox-init launch.  Every upload below is followed by a stream
  // synchronise before this function returns: the buffer is free again for the next call of this thread.
  const size_t o_bbox = align_up(sizeof(int32_t) * (batch + 1), 256);
  const size_t o_blk = o_bbox + align_up(sizeof(uint32_t) * 6 * batch, 256);
  const size_t o_tick = o_blk + align_up(sizeof(int32_t) * (batch + 1), 256);
  const size_t o_hb = o_tick + 256;
  const size_t o_grids = o_hb + align_up(sizeof(uint32_t) * 6 * batch, 256);
  const size_t o_cblk = o_grids + align_up(sizeof(CloudGrid) * batch, 256);  // (the carver's alignment: one upload for both)
  const size_t o_mb = o_cblk + align_up(sizeof(int32_t) * (batch + 1), 256);
  char* pin = static_cast<char*>(pinned_scratch(7, o_mb + sizeof(int32_t) * (batch + 2)));
  GR_REQUIRE(pin != nullptr, "grid_subsample: pinned staging buffer could not be allocated");
  GR_REQUIRE(reinterpret_cast<char*>(w.bbox) == reinterpret_cast<char*>(w.off) + o_bbox &&
                 reinterpret_cast<char*>(w.blk_off) == reinterpret_cast<char*>(w.off) + o_blk &&
                 reinterpret_cast<char*>(w.ticket) == reinterpret_cast<char*>(w.off) + o_tick,
             "grid_subsample: workspace layout");
  int32_t* off = reinterpret_cast<int32_t*>(pin);
  uint32_t* bb0 = reinterpret_cast<uint32_t*>(pin + o_bbox);
  int32_t* h_blk = reinterpret_cast<int32_t*>(pin + o_blk);
  const uint32_t* hb = reinterpret_cast<const uint32_t*>(pin + o_hb);
  *reinterpret_cast<int32_t*>(pin + o_tick) = 0;
  // small batches: the two read-backs (bounding boxes, cell counts) come through the mailbox page -- the kernels post
  // them, the host polls: no copy in the stream, no stream synchronise (words [0, 6 B] boxes + stamp, [512, 512 + B + 1]
  // counts + stamp)
  // (the boxes of up to 80 clouds fit their half of the page; the counts of up to 256 -- cloud_counts_kernel posts from a
  // single workgroup -- fit theirs: a 64-pair pyramid level, 128 clouds, waits for its counts without a copy + synchronise)
  volatile int32_t* mail = batch <= 256 ? mailbox() : nullptr;
  const int stamp = mail ? mailbox_next_stamp() : 0;
  CloudGrid* hg = reinterpret_cast<CloudGrid*>(pin + o_grids);
  const int32_t* h_mb = reinterpret_cast<const int32_t*>(pin + o_mb);  // (or the mailbox page, below)
  off[0] = 0;
  for (int64_t b = 0; b < batch; ++b) {
    off[b + 1] = off[b] + (int32_t)h_lengths[b];
    for (int k = 0; k < 6; ++k) bb0[b * 6 + k] = k < 3 ? 0xffffffffu : 0u;
  }
  bbox_block_offsets(off, h_blk, nb);
  GR_HIP(hipMemcpyAsync(w.off, pin, o_tick + sizeof(int32_t), hipMemcpyHostToDevice, stream));
  // (the ticket is one same-address atomic + a fence per workgroup: fine for a few hundred workgroups, 1.4 ms for the
  // 12 500 of a 64 x 200 k call -- those copy and synchronise)
  const bool bbox_by_mail = mail != nullptr && batch <= 80 && h_blk[nb] <= 512;
  if (bbox_by_mail) mailbox_arm(mail + MAIL_GRID_BOXES + 6 * batch);
  int rc = compute_bbox(points, off, h_blk, w.off, nb, w.bbox, w.blk_off, stream, /*blk_off_on_device=*/true,
                        /*init_bbox=*/false, bbox_by_mail ? w.ticket : nullptr, const_cast<int32_t*>(mail), stamp);
  if (rc != GR_OK) {
    (void)hipStreamSynchronize(stream);  // nothing queued may post into the page after this call has returned
    return rc;
  }
  if (bbox_by_mail) {  // (n > 0 here: at least one bbox workgroup runs and posts)
    rc = mailbox_wait(mail + 6 * batch, stamp, stream, "grid_subsample (bounding boxes)");
    if (rc != GR_OK) return rc;
    hb = const_cast<const uint32_t*>(reinterpret_cast<const volatile uint32_t*>(mail));
  } else {
    GR_HIP(hipMemcpyAsync(pin + o_hb, w.bbox, sizeof(uint32_t) * batch * 6, hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
  }

  // ---- per-cloud grid, exactly the reference's fp32 expressions (this TU: -ffp-contract=off)
  unsigned long long max_cells = 1;
  bool wrap = false;
  int max_dig = 3, max_dig_fo = 3;
  constexpr int64_t MIN_BUCKETS = 8192;  // (64 pairs of 30 000-point clouds: 6 bits 0.247 ms per call, 7: 0.256, 8: 0.272, 9: 0.288)
  const float inv = static_cast<float>(1.0 / static_cast<double>(voxel));  // :11 `(1. / voxel_size)` -> float
  for (int64_t b = 0; b < batch; ++b) {
    CloudGrid g{0.f, 0.f, 0.f, voxel, 1ull, 1ull, 0, 0};
    if (h_lengths[b] > 0) {
      const float mnx = ord2f(hb[b * 6 + 0]), mny = ord2f(hb[b * 6 + 1]), mnz = ord2f(hb[b * 6 + 2]);
      const float mxx = ord2f(hb[b * 6 + 3]), mxy = ord2f(hb[b * 6 + 4]), mxz = ord2f(hb[b * 6 + 5]);
      g.ox = std::floor(mnx * inv) * voxel;
      g.oy = std::floor(mny * inv) * voxel;
      g.oz = std::floor(mnz * inv) * voxel;
      const unsigned long long NX = to_size_t(std::floor((double)((mxx - g.ox) / voxel)) + 1);
      const unsigned long long NY = to_size_t(std::floor((double)((mxy - g.oy) / voxel)) + 1);
      const unsigned long long NZ = to_size_t(std::floor((double)((mxz - g.oz) / voxel)) + 1);
      g.nx = NX;
      g.nxny = NX * NY;
      // can the smallest coordinate land in a negative cell (origin rounded above the min)?
      if ((mnx - g.ox) / voxel < 0.f || (mny - g.oy) / voxel < 0.f || (mnz - g.oz) / voxel < 0.f) wrap = true;
      const long double cells = (long double)NX * (long double)NY * (long double)NZ;
      if (cells >= 18446744073709551615.0L || !std::isfinite((double)cells)) wrap = true;
      else {
        if ((unsigned long long)cells > max_cells) max_cells = (unsigned long long)cells;
        // top digit of the bucket sorts: nine bits for a large cloud; a small one takes fewer, so that a bucket still holds a
        // few hundred points (30 000 points over 512 buckets were 58 per 128-thread workgroup: 65 000 workgroups of fixed
        // costs per 64-pair level, 77 us of a 0.29 ms call).  What is left of a key has to fit the 18 bits of an LDS item.
        const int len_bits = bits_for((unsigned long long)h_lengths[b]), cell_bits = bits_for((unsigned long long)cells);
        // (only where the batch still fills the device with the fewer, larger buckets: two 30 000-point clouds on 2 x 64
        // buckets measured 0.092 ms against 0.081 on 2 x 512)
        int dig_small = std::min(std::max(len_bits - 9, 3), 9);
        while (dig_small < 9 && (batch << dig_small) < MIN_BUCKETS) ++dig_small;
        const int dig = std::max(dig_small, cell_bits - 18);
        const int dig_fo = dig_small;
        g.sh = std::max(0, cell_bits - dig);    // fields 1 .. cells: (cells - 1) >> sh < 2^dig
        g.sh_fo = std::max(0, len_bits - dig_fo);
        max_dig = std::max(max_dig, std::min(dig, 9));
        max_dig_fo = std::max(max_dig_fo, dig_fo);
      }
    }
    hg[b] = g;
  }
  int32_t* h_cblk = reinterpret_cast<int32_t*>(pin + o_cblk);
  h_cblk[0] = 0;
  for (int64_t b = 0; b < batch; ++b) h_cblk[b + 1] = h_cblk[b] + (int32_t)((h_lengths[b] + 255) / 256);
  GR_REQUIRE(reinterpret_cast<char*>(w.cblk) == reinterpret_cast<char*>(w.grids) + (o_cblk - o_grids), "grid_subsample: workspace layout");
  GR_HIP(hipMemcpyAsync(w.grids, hg, (o_cblk - o_grids) + sizeof(int32_t) * (batch + 1), hipMemcpyHostToDevice, stream));
  int key_bits = wrap ? 64 : bits_for(max_cells);
  const int b_bits = bits_for((unsigned long long)batch);
  const bool composite = key_bits + b_bits <= 64 && key_bits < 64;
  if (!composite) key_bits = 64;

  const dim3 blk(256), grd((unsigned)((n + 255) / 256));
  uint64_t* keys_sorted = w.keys_b;
  int32_t* vals_sorted = w.vals_b;
  // Bucket sort (depth_sort.hip: ONE pass over the top nine bits of every cloud's own key range, then every bucket -- a few
  // hundred points, whole voxels -- is finished inside LDS): two trips through memory instead of three radix passes with
  // three launches each.  For voxel keys of <= 26 bits, clouds of <= 2^20 points and a batch that is not too ragged (the
  // tables are laid out for the longest cloud); a bucket that outgrows LDS (a cloud crowded into a few voxel slabs) raises
  // a flag that comes back with the cell counts, and the call starts over with the general sort.
  int64_t max_len = 0;
  for (int64_t b = 0; b < batch; ++b) max_len = std::max<int64_t>(max_len, h_lengths[b]);
  const bool bucket = allow_bucket_sort && composite && !wrap && key_bits <= 26 && max_len <= (1ll << 20) &&
                      batch * ((max_len + 2047) / 2048) <= w.ds_rows && depth_sort_table_bytes(max_len, nb) <= w.ds_bytes;
  // (reference order) rank the cells by their first point with a second bucket sort instead of flags + a scan over all points:
  // five launches against four, so only where launches are not the cost (one 200 k cloud: 0.194 ms with the flags, 0.207 with
  // the sort; 64 of them: 1.26 -> 1.17 ms)
  const bool fo_by_sort = bucket && order_mode != GR_ORDER_CELL && n >= (1ll << 21);
  if (bucket) {
    uint32_t* field = reinterpret_cast<uint32_t*>(w.keys_a);
    uint32_t* payload = field + n;
    hipLaunchKernelGGL(keys32_kernel, grd, blk, 0, stream, points, (int)n, w.off, nb, w.grids, field, payload,
                       order_mode != GR_ORDER_CELL && !fo_by_sort ? w.flags : nullptr, w.ds_range, w.ds_ovf, w.ds_range_fo);
    GR_LAUNCH_CHECK();
    const DepthSortSegments sg{w.off, w.ds_range, w.keys_fo, key_bits, nullptr, nullptr, nullptr, 1 << max_dig};
    rc = depth_sort_views(field, payload, w.keys_b, w.keys_b, w.vals_b, reinterpret_cast<uint32_t*>(w.scan), w.ds_nvalid, max_len, nb,
                          27, w.ds_table, w.ds_bytes, stream, nullptr, 0, w.ds_ovf, 1, nullptr, &sg);
    if (rc != GR_OK) return rc;
    keys_sorted = w.keys_fo;  // (cloud << key_bits | voxel key) words, as the general sort leaves them
  } else {
  hipLaunchKernelGGL(keys_kernel, grd, blk, 0, stream, points, (int)n, w.off, nb, w.grids, key_bits, w.keys_a, w.vals_a,
                     order_mode != GR_ORDER_CELL ? w.flags : nullptr);
  GR_LAUNCH_CHECK();
  if (composite || batch == 1) {
    rc = sort_pairs_u64_i32(w.sort_temp, w.sort_temp_bytes, w.keys_a, w.keys_b, w.vals_a, w.vals_b, n, 0,
                            composite ? key_bits + b_bits : 64, stream);
    if (rc != GR_OK) return rc;
  } else {
    // keys need all 64 bits: stable sort by key, then stable sort by cloud id
    rc = sort_pairs_u64_i32(w.sort_temp, w.sort_temp_bytes, w.keys_a, w.keys_b, w.vals_a, w.vals_b, n, 0, 64, stream);
    if (rc != GR_OK) return rc;
    hipLaunchKernelGGL(batch_keys_kernel, grd, blk, 0, stream, w.vals_b, (int)n, w.off, nb, w.keys_a);
    rc = sort_pairs_u64_i32(w.sort_temp, w.sort_temp_bytes, w.keys_a, w.keys_fo, w.vals_b, w.vals_a, n, 0,
                            b_bits, stream);
    if (rc != GR_OK) return rc;
    // vals_a now holds point indices in (cloud, key, index) order; recompute their raw keys
    vals_sorted = w.vals_a;
    keys_sorted = w.keys_b;  // overwritten by regather_keys
    hipLaunchKernelGGL(regather_keys_kernel, grd, blk, 0, stream, points, vals_sorted, (int)n, w.off, nb, w.grids, keys_sorted);
  }
  }
  GR_LAUNCH_CHECK();

  // ---- runs -> cells
  int32_t* blk_cnt = w.scan;    // run heads per workgroup of 256 sorted positions
  int32_t* blk_base = w.perm;   // ... and their exclusive scan (perm is free until the very end)
  const int64_t nblk = h_cblk[nb];  // (>= 1: n > 0)
  hipLaunchKernelGGL(head_count_kernel, dim3((unsigned)nblk), blk, 0, stream, keys_sorted, vals_sorted, (int)n, w.off, nb,
                     composite ? 1 : 0, w.cblk, blk_cnt);
  rc = exclusive_scan_i32(blk_cnt, blk_base, nblk, 1, nblk, w.scan_ws, w.totals, stream);
  if (rc != GR_OK) return rc;
  // reference order: the cells are ranked by their first point.  General path: a flag per point (zeroed by keys_kernel, set by
  // cells_kernel), a scan over all points, a launch that reads every cell's rank off it.  Bucket path: cells_kernel leaves
  // (first point inside its cloud) + 1 per cell and the bucket sort orders the cells of every cloud by it -- sorted position =
  // rank, so the ids it writes ARE cell_of_rank and its 64-bit output gathers the cells' keys in that order.
  const bool fo_sort = fo_by_sort;
  int32_t* fo_flags = order_mode == GR_ORDER_CELL || fo_sort ? nullptr : w.flags;  // zeroed by keys_kernel
  // cell order: the barycentres ARE the output rows (cell = rank of the voxel key), written in place
  hipLaunchKernelGGL(cells_kernel, dim3((unsigned)((nblk + 7) / 8 * 8)), blk, 0, stream, points, keys_sorted, vals_sorted, blk_base,
                     composite ? 1 : 0, w.cblk, (int)nblk, (int)n,
                     w.off, nb, composite ? key_bits : 64, order_mode == GR_ORDER_CELL ? out_points : w.bary, w.first_idx,
                     w.cell_key, w.cell_batch, fo_flags, fo_sort ? 1 : 0);
  int32_t* mail_counts = mail ? const_cast<int32_t*>(mail) + MAIL_GRID_COUNTS : nullptr;  // (nb <= 80: one workgroup)
  const int stamp2 = mail ? mailbox_next_stamp() : 0;
  if (mail) mailbox_arm(mail + MAIL_GRID_COUNTS + batch + 2);
  hipLaunchKernelGGL(cloud_counts_kernel, dim3((nb + 255) / 256), blk, 0, stream, w.cblk, (int)nblk, blk_base, w.totals, (int)n, w.off,
                     nb, w.m_b, mail_counts, stamp2, bucket ? w.ds_ovf : nullptr, w.cell_off);
  GR_LAUNCH_CHECK();
  if (fo_sort) {
    // (no bucket of this sort can overflow: its digit is the top nine bits of a point index, so a bucket holds the cells whose
    // first point lies in a span of <= len / 512 <= 2 048 points)
    uint32_t* fo_field = reinterpret_cast<uint32_t*>(w.first_idx);  // also the (unused, < 2^26) payload
    const DepthSortSegments sg2{w.cell_off, w.ds_range_fo, w.keys_fo, 0, nullptr, nullptr, w.cell_key, 1 << max_dig_fo};
    rc = depth_sort_views(fo_field, fo_field, w.keys_b, w.keys_b, w.cell_of_rank, reinterpret_cast<uint32_t*>(w.scan), w.ds_nvalid,
                          max_len, nb, 27, w.ds_table, w.ds_bytes, stream, nullptr, 0, w.ds_ovf, 1, nullptr, &sg2);
    if (rc != GR_OK) return rc;
  }
  int32_t h_m = 0;
  // (bucket sort) a bucket overflowed: nothing behind the sort is valid -- the whole call again, general sort
  auto start_over = [&]() -> int {
    GR_HIP(hipStreamSynchronize(stream));  // nothing of this attempt may still post into the mailbox page
    g_grid_bucket_fallbacks.fetch_add(1);
    return grid_subsample_impl(points, h_lengths, n, batch, voxel, order_mode, out_points, h_out_lengths, h_total_m, ws,
                               ws_bytes, stream_, false);
  };
  if (mail) h_mb = const_cast<const int32_t*>(reinterpret_cast<const volatile int32_t*>(mail + MAIL_GRID_COUNTS));
  else GR_HIP(hipMemcpyAsync(pin + o_mb, w.m_b, sizeof(int32_t) * (batch + 2), hipMemcpyDeviceToHost, stream));
  // (mailbox: the counts are on the host as soon as cloud_counts_kernel has run -- for the reference order that is while
  // the first-occurrence scan below is still running)
  auto wait_counts = [&]() -> int {
    if (mail) return mailbox_wait(mail + MAIL_GRID_COUNTS + batch + 2, stamp2, stream, "grid_subsample (cell counts)");
    GR_HIP(hipStreamSynchronize(stream));
    return GR_OK;
  };

  if (order_mode == GR_ORDER_CELL) {
    rc = wait_counts();
    if (rc != GR_OK) return rc;
    if (bucket && h_mb[batch + 1] != 0) return start_over();
    h_m = h_mb[batch];
  } else {
    // first-occurrence rank of every cell, keys in that order -> host
    int32_t* fo_scan = w.scan;  // head flags no longer needed
    if (!fo_sort) {
      rc = exclusive_scan_i32(fo_flags, fo_scan, n, 1, n, w.scan_ws, w.totals + 1, stream);
      if (rc != GR_OK) return rc;
    }
    rc = wait_counts();
    if (rc != GR_OK) return rc;
    if (bucket && h_mb[batch + 1] != 0) return start_over();
    h_m = h_mb[batch];
    if (h_m > 0) {
      if (!fo_sort)
      hipLaunchKernelGGL(fo_rank_kernel, dim3((unsigned)(((h_m + 255) / 256 + 7) / 8 * 8)), blk, 0, stream, w.first_idx, w.cell_key,
                         fo_scan, h_m, w.cell_of_rank, w.keys_fo);
      GR_LAUNCH_CHECK();
      // The reference inserts keys in first-occurrence order into an unordered_map and emits in its iteration order
      // (grid_subsampling_cpu.cpp:28-47).  Default: that order is evaluated on the device in closed form
      // (hash_order_device.hip; the bucket counts come from the real libstdc++ rehash policy).  The host replay of the
      // container's linking rules (hash_order.hip, gr_host_unordered_map_order) is its checker: tests/test_gpu_hash_order.py.
      std::vector<int64_t> r0(batch + 1, 0);
      for (int64_t b = 0; b < batch; ++b) r0[b + 1] = r0[b] + h_mb[b];
      // the last launch of the evaluation moves the barycentres to their rows itself (no permutation, no gather launch)
      rc = hash_order_device(w.keys_fo, r0.data(), batch, nullptr, w.ho_ws, w.ho_bytes, stream, w.bary, w.cell_of_rank,
                             out_points);
      if (rc != GR_OK) return rc;
      GR_LAUNCH_CHECK();
    }
  }
  for (int64_t b = 0; b < batch; ++b) h_out_lengths[b] = h_mb[b];
  *h_total_m = h_m;
  return GR_OK;
}

extern "C" int gr_grid_subsample(const float* points, const int64_t* h_lengths, int64_t n,
                                 int64_t batch, float voxel, int order_mode, float* out_points,
                                 int64_t* h_out_lengths, int64_t* h_total_m, void* ws,
                                 size_t ws_bytes, void* stream_) {
  return grid_subsample_impl(points, h_lengths, n, batch, voxel, order_mode, out_points, h_out_lengths, h_total_m, ws, ws_bytes,
                             stream_, gr::g_grid_bucket_sort.load() != 0);
}

extern "C" int gr_grid_subsample_debug_bucket_sort(int on) {
  if (on == 2) return gr::g_grid_bucket_fallbacks.load();
  const int old = gr::g_grid_bucket_sort.load();
  if (on == 0 || on == 1) gr::g_grid_bucket_sort.store(on);
  return old;
}
