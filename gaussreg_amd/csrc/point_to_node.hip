// point_to_node_partition on MI355X without the (M, N) distance matrix.
//
// Replaces  geotransformer/modules/ops/pointcloud_partition.py:61-111  (a11), which materialises
// sq_dist_mat (M,N) fp32 + a bool mask (M,N) and runs argmin + masked topk over them.  Here:
//   assign   one thread per point, nodes staged in LDS: argmin_m d(m,n) with the reference's
//            expanded form d = clamp((|node|^2 - 2 node.point) + |point|^2, 0)  (pairwise_distance.py:28-31)
//   group    counting sort of the points by owning node (histogram, scan, scatter)
//   select   one workgroup per node: its points' (d, index) keys bitonic-sorted in LDS, the
//            point_limit smallest emitted in ascending order, the rest padded with N / False
// Algorithmic traffic: 12(N+M) + 8N + M(1 + 9K) bytes instead of >= 5 N M bytes.
#include <algorithm>

#include "common.hpp"

namespace gr {
namespace {

__device__ __forceinline__ float node_point_dist(float4 nd, float px, float py, float pz, float p2) {
  const float xy = fmaf(nd.z, pz, fmaf(nd.y, py, nd.x * px));  // 3-term dot product
  return fmaxf((nd.w - 2.0f * xy) + p2, 0.0f);
}

constexpr int P2N_CHUNK = 1024;  // nodes staged per LDS round

__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ pts, int n,
                                                     const float* __restrict__ nodes, int m,
                                                     int64_t* __restrict__ point_to_node,
                                                     int32_t* __restrict__ owner, float* __restrict__ owner_d,
                                                     int32_t* __restrict__ node_cnt, uint8_t* __restrict__ node_masks,
                                                     const int32_t* __restrict__ point_off,
                                                     const int32_t* __restrict__ node_off) {
  __shared__ float4 s_nodes[P2N_CHUNK];
  // stack mode: blockIdx.y = cloud; its points may only go to its own nodes.  `owner` and the node tables use GLOBAL node
  // ids (node_off[cloud] + local id), point_to_node the local id the reference returns.
  int pbase = 0, nbase = 0;
  if (point_off) {
    pbase = point_off[blockIdx.y];
    n = point_off[blockIdx.y + 1] - pbase;
    nbase = node_off[blockIdx.y];
    m = node_off[blockIdx.y + 1] - nbase;
    if ((int)blockIdx.x * 256 >= n) return;  // the grid is sized for the largest cloud
    pts += 3 * (int64_t)pbase;
    nodes += 3 * (int64_t)nbase;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f, p2 = 0.f;
  if (i < n) {
    px = pts[3 * (int64_t)i];
    py = pts[3 * (int64_t)i + 1];
    pz = pts[3 * (int64_t)i + 2];
    p2 = (px * px + py * py) + pz * pz;
  }
  float best = INFINITY;
  int bm = 0;
  for (int m0 = 0; m0 < m; m0 += P2N_CHUNK) {
    const int cnt = min(P2N_CHUNK, m - m0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const float x = nodes[3 * (int64_t)(m0 + j)], y = nodes[3 * (int64_t)(m0 + j) + 1], z = nodes[3 * (int64_t)(m0 + j) + 2];
      s_nodes[j] = make_float4(x, y, z, (x * x + y * y) + z * z);
    }
    __syncthreads();
    if (i < n) {
      for (int j = 0; j < cnt; ++j) {
        const float d = node_point_dist(s_nodes[j], px, py, pz, p2);
        if (d < best) {  // strict: first minimum wins, like torch.min
          best = d;
          bm = m0 + j;
        }
      }
    }
  }
  if (i < n && m == 0) owner[pbase + i] = -1;  // (stack mode only: a cloud without nodes)
  if (i < n && m > 0) {
    point_to_node[pbase + i] = bm;
    owner[pbase + i] = nbase + bm;
    owner_d[pbase + i] = best;
    atomicAdd(&node_cnt[nbase + bm], 1);
    node_masks[nbase + bm] = 1;  // pointcloud_partition.py:88-89 index_fill_(True)
  }
}

__global__ __launch_bounds__(256) void scatter_points_kernel(int n, const int32_t* __restrict__ owner,
                                                             const float* __restrict__ owner_d,
                                                             const int32_t* __restrict__ node_start,
                                                             int32_t* __restrict__ node_cnt,
                                                             unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int m = owner[i];
  if (m < 0) return;
  const int slot = node_start[m] + atomicSub(&node_cnt[m], 1) - 1;
  keys[slot] = ((unsigned long long)__float_as_uint(owner_d[i]) << 32) | (unsigned int)i;  // d >= 0: bits are monotone
}

constexpr int SEL_N = 2048;  // LDS sort width
constexpr unsigned long long KEY_INF = ~0ull;

// one block per node: K smallest (d, index) keys, ascending
__global__ __launch_bounds__(256) void select_kernel(int n, int K, const int32_t* __restrict__ node_start,
                                                     const unsigned long long* __restrict__ keys,
                                                     int64_t* __restrict__ knn_idx, uint8_t* __restrict__ knn_mask,
                                                     const int32_t* __restrict__ point_off,
                                                     const int32_t* __restrict__ node_off, int nclouds) {
  __shared__ unsigned long long sk[SEL_N];
  const int m = blockIdx.x;
  int pbase = 0;  // stack mode: keys carry global point indices; the output is local to the node's cloud
  if (point_off) {
    const int c = find_batch(node_off, nclouds, m);
    pbase = point_off[c];
    n = point_off[c + 1] - pbase;
  }
  const int a = node_start[m], b = node_start[m + 1];
  if (b - a <= WAVE) {
    // A node with at most 64 points (nearly every node of GaussReg's pyramid: 9 000 fine points over 770 nodes): one wave
    // sorts them in registers, one key per lane (bitonic network over ds_bpermute; 21 exchange steps), no LDS, no barrier.
    // The 2048-wide LDS sort below costs 66 barrier-separated passes whatever the node holds: 2.1 ms per 98 000 nodes.
    if (threadIdx.x >= WAVE) return;
    const int lane = threadIdx.x;
    unsigned long long key = lane < b - a ? keys[a + lane] : KEY_INF;
#pragma unroll
    for (int size = 2; size <= WAVE; size <<= 1) {
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        const unsigned lo_o = (unsigned)__shfl_xor((int)(unsigned)key, stride, WAVE);
        const unsigned hi_o = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), stride, WAVE);
        const unsigned long long other = ((unsigned long long)hi_o << 32) | lo_o;
        const bool asc = (lane & size) == 0, lower = (lane & stride) == 0;
        const bool take_min = lower == asc;
        key = (other < key) == take_min ? other : key;
      }
    }
    // lane j holds the j-th smallest key; outputs past the node's points are padding
    for (int j = lane; j < K; j += WAVE) {
      const unsigned long long kj = j < WAVE ? key : KEY_INF;  // (j < 64 only in the first trip, where j == lane)
      const bool ok = kj != KEY_INF;
      knn_idx[(int64_t)m * K + j] = ok ? (int64_t)((unsigned int)(kj & 0xffffffffull) - (unsigned)pbase) : (int64_t)n;
      knn_mask[(int64_t)m * K + j] = ok ? 1 : 0;
    }
    return;
  }
  // best-K kept in sk[0..K); each round appends up to SEL_N - K fresh keys and re-sorts
  for (int i = threadIdx.x; i < SEL_N; i += 256) sk[i] = KEY_INF;
  __syncthreads();
  int pos = a;
  bool first = true;
  do {
    const int keep = first ? 0 : K;
    const int take = min(SEL_N - keep, b - pos);
    for (int i = threadIdx.x; i < SEL_N - keep; i += 256) sk[keep + i] = i < take ? keys[pos + i] : KEY_INF;
    pos += take;
    first = false;
    __syncthreads();
    for (int size = 2; size <= SEL_N; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = threadIdx.x; t < SEL_N / 2; t += 256) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool asc = (lo & size) == 0;
          const unsigned long long x = sk[lo], y = sk[hi];
          if ((x > y) == asc) {
            sk[lo] = y;
            sk[hi] = x;
          }
        }
        __syncthreads();
      }
    }
  } while (pos < b);
  for (int j = threadIdx.x; j < K; j += 256) {
    const unsigned long long key = sk[j];
    const bool ok = key != KEY_INF;
    knn_idx[(int64_t)m * K + j] = ok ? (int64_t)((unsigned int)(key & 0xffffffffull) - (unsigned)pbase) : (int64_t)n;  // :102 pad = N
    knn_mask[(int64_t)m * K + j] = ok ? 1 : 0;                                                      // :101
  }
}

struct P2nWs {
  int32_t* owner;
  float* owner_d;
  int32_t* node_cnt;
  int32_t* node_start;
  int32_t* scan_ws;
  unsigned long long* keys;
  size_t bytes;
};

P2nWs carve_p2n(void* p, int64_t n, int64_t m) {
  P2nWs w;
  Carver c(p);
  w.owner = c.take<int32_t>(n);
  w.owner_d = c.take<float>(n);
  w.node_cnt = c.take<int32_t>(m + 1);
  w.node_start = c.take<int32_t>(m + 1);
  w.scan_ws = c.take<int32_t>(scan_ws_ints(m + 1));
  w.keys = c.take<unsigned long long>(n);
  w.bytes = c.used();
  return w;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_point_to_node_workspace_bytes(int64_t n, int64_t m) {
  if (n < 0 || m < 0) return 0;
  return carve_p2n(nullptr, n, m).bytes;
}

extern "C" int gr_point_to_node_partition(const float* points, int64_t n, const float* nodes, int64_t m,
                                          int point_limit, int64_t* point_to_node, uint8_t* node_masks,
                                          int64_t* node_knn_indices, uint8_t* node_knn_masks, void* ws,
                                          size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && m >= 0 && n < (1ll << 31) - 1 && m < (1ll << 31) - 1, "bad sizes");
  GR_REQUIRE(point_limit >= 1 && point_limit <= SEL_N / 2, "point_limit must be in [1, %d]", SEL_N / 2);
  GR_REQUIRE(n >= point_limit, "need at least point_limit points (torch.topk would fail too)");
  if (m == 0) return GR_OK;
  GR_REQUIRE(points && nodes && point_to_node && node_masks && node_knn_indices && node_knn_masks, "null argument");
  P2nWs w = carve_p2n(ws, n, m);
  if (!ws || ws_bytes < w.bytes) {
    set_error("point_to_node workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes);
    return GR_ERR_WORKSPACE;
  }
  GR_HIP(hipMemsetAsync(w.node_cnt, 0, sizeof(int32_t) * (m + 1), stream));
  GR_HIP(hipMemsetAsync(node_masks, 0, (size_t)m, stream));
  hipLaunchKernelGGL(assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points, (int)n, nodes, (int)m,
                     point_to_node, w.owner, w.owner_d, w.node_cnt, node_masks, (const int32_t*)nullptr, (const int32_t*)nullptr);
  GR_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(w.node_cnt, w.node_start, m + 1, 1, m + 1, w.scan_ws, nullptr, stream);
  if (rc != GR_OK) return rc;
  hipLaunchKernelGGL(scatter_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (int)n, w.owner,
                     w.owner_d, w.node_start, w.node_cnt, w.keys);
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)m), dim3(256), 0, stream, (int)n, point_limit, w.node_start, w.keys,
                     node_knn_indices, node_knn_masks, (const int32_t*)nullptr, (const int32_t*)nullptr, 0);
  GR_LAUNCH_CHECK();
  return GR_OK;
}

// Stack mode over `nclouds` (fine cloud, its coarse nodes) pairs -- every cloud of a batch of scene pairs in one call
// (model.py:99-104 runs the single-cloud form twice per pair).  The batch is ONE problem for the kernels above: a point may
// only be assigned to the nodes of its own cloud (grid.y = cloud), the node histogram / scan / scatter run over the stacked
// node list, every node's selection finds its cloud in the offset table.  Six stream operations for the whole batch, no host
// synchronisation.  Outputs are the single-cloud outputs concatenated; indices stay LOCAL to their cloud (padding value =
// that cloud's point count).
static size_t p2n_batch_bytes(int64_t total_n, int64_t total_m, int64_t nclouds) {
  return align_up(carve_p2n(nullptr, total_n, total_m).bytes, 256) + 2 * align_up((size_t)(nclouds + 1) * sizeof(int32_t), 256);
}

extern "C" size_t gr_point_to_node_batch_workspace_bytes(const int64_t* h_point_off, const int64_t* h_node_off,
                                                         int64_t nclouds) {
  if (!h_point_off || !h_node_off || nclouds < 0) return 0;
  return p2n_batch_bytes(h_point_off[nclouds] - h_point_off[0], h_node_off[nclouds] - h_node_off[0], nclouds);
}

extern "C" int gr_point_to_node_partition_batch(const float* points, const int64_t* h_point_off, const float* nodes,
                                                const int64_t* h_node_off, int64_t nclouds, int point_limit,
                                                int64_t* point_to_node, uint8_t* node_masks, int64_t* node_knn_indices,
                                                uint8_t* node_knn_masks, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(nclouds >= 0 && nclouds < 65536 && h_point_off && h_node_off, "bad arguments");
  GR_REQUIRE(point_limit >= 1 && point_limit <= SEL_N / 2, "point_limit must be in [1, %d]", SEL_N / 2);
  GR_REQUIRE(h_point_off[0] == 0 && h_node_off[0] == 0, "offsets start at 0");
  if (nclouds == 0) return GR_OK;
  int64_t max_n = 0;
  for (int64_t c = 0; c < nclouds; ++c) {
    const int64_t n = h_point_off[c + 1] - h_point_off[c], m = h_node_off[c + 1] - h_node_off[c];
    GR_REQUIRE(n >= 0 && m >= 0, "cloud %lld: offsets must ascend", (long long)c);
    GR_REQUIRE(m == 0 || n >= point_limit, "cloud %lld: need at least point_limit points (torch.topk would fail too)", (long long)c);
    max_n = std::max(max_n, n);
  }
  const int64_t total_n = h_point_off[nclouds], total_m = h_node_off[nclouds];
  GR_REQUIRE(total_n < (1ll << 31) - 1 && total_m < (1ll << 31) - 1, "bad sizes");
  if (total_m == 0) return GR_OK;
  GR_REQUIRE(points && nodes && point_to_node && node_masks && node_knn_indices && node_knn_masks, "null argument");
  if (!ws || ws_bytes < p2n_batch_bytes(total_n, total_m, nclouds)) {
    set_error("point_to_node workspace too small");
    return GR_ERR_WORKSPACE;
  }
  P2nWs w = carve_p2n(ws, total_n, total_m);
  char* tail = static_cast<char*>(ws) + align_up(w.bytes, 256);
  int32_t* d_po = reinterpret_cast<int32_t*>(tail);
  int32_t* d_no = reinterpret_cast<int32_t*>(tail + align_up((size_t)(nclouds + 1) * sizeof(int32_t), 256));
  // the two offset tables go up through pinned per-thread staging; an event says when the copies have left it
  static thread_local hipEvent_t staged = nullptr;
  if (staged == nullptr) GR_HIP(hipEventCreateWithFlags(&staged, hipEventDisableTiming));
  else GR_HIP(hipEventSynchronize(staged));
  int32_t* stage = static_cast<int32_t*>(pinned_scratch(5, 2 * sizeof(int32_t) * (nclouds + 1)));
  GR_REQUIRE(stage != nullptr, "pinned staging buffer could not be allocated");
  for (int64_t c = 0; c <= nclouds; ++c) stage[c] = (int32_t)h_point_off[c], stage[nclouds + 1 + c] = (int32_t)h_node_off[c];
  GR_HIP(hipMemcpyAsync(d_po, stage, sizeof(int32_t) * (nclouds + 1), hipMemcpyHostToDevice, stream));
  GR_HIP(hipMemcpyAsync(d_no, stage + nclouds + 1, sizeof(int32_t) * (nclouds + 1), hipMemcpyHostToDevice, stream));
  GR_HIP(hipEventRecord(staged, stream));
  GR_HIP(hipMemsetAsync(w.node_cnt, 0, sizeof(int32_t) * (total_m + 1), stream));
  GR_HIP(hipMemsetAsync(node_masks, 0, (size_t)total_m, stream));
  hipLaunchKernelGGL(assign_kernel, dim3((unsigned)((max_n + 255) / 256), (unsigned)nclouds), dim3(256), 0, stream, points, 0,
                     nodes, 0, point_to_node, w.owner, w.owner_d, w.node_cnt, node_masks, d_po, d_no);
  GR_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(w.node_cnt, w.node_start, total_m + 1, 1, total_m + 1, w.scan_ws, nullptr, stream);
  if (rc != GR_OK) return rc;
  hipLaunchKernelGGL(scatter_points_kernel, dim3((unsigned)((total_n + 255) / 256)), dim3(256), 0, stream, (int)total_n,
                     w.owner, w.owner_d, w.node_start, w.node_cnt, w.keys);
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)total_m), dim3(256), 0, stream, 0, point_limit, w.node_start, w.keys,
                     node_knn_indices, node_knn_masks, d_po, d_no, (int)nclouds);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
