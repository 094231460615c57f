// Error plumbing + a small multi-row exclusive scan used by the binning kernels.
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"

namespace gr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------ kernel timing registry
namespace {
struct TimingRec {
  std::string name;
  hipEvent_t start, stop;
};
std::mutex g_tmutex;
std::vector<TimingRec> g_trecs;
bool g_timing_on = false;
}  // namespace

KernelTimer::KernelTimer(const char* name, hipStream_t stream) : slot_(-1), stream_(stream) {
  if (!g_timing_on) return;
  std::lock_guard<std::mutex> lk(g_tmutex);
  TimingRec r;
  r.name = name;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  (void)hipEventRecord(r.start, stream);
  g_trecs.push_back(r);
  slot_ = (int)g_trecs.size() - 1;
}

KernelTimer::~KernelTimer() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lk(g_tmutex);
  (void)hipEventRecord(g_trecs[slot_].stop, stream_);
}

// ------------------------------------------------------------------ pinned scratch
void* pinned_scratch(int slot, size_t bytes) {
  constexpr int SLOTS = 10;
  static thread_local void* buf[SLOTS] = {};
  static thread_local size_t cap[SLOTS] = {};
  if (slot < 0 || slot >= SLOTS) return nullptr;
  if (cap[slot] < bytes) {
    if (buf[slot]) (void)hipHostFree(buf[slot]);
    buf[slot] = nullptr;
    cap[slot] = 0;
    const size_t want = bytes < 4096 ? 4096 : 2 * bytes;
    if (hipHostMalloc(&buf[slot], want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      buf[slot] = nullptr;
      return nullptr;
    }
    cap[slot] = want;
  }
  return buf[slot];
}

// ------------------------------------------------------------------ host mailbox
volatile int32_t* mailbox() {
  static thread_local int32_t* page = nullptr;
  static thread_local bool tried = false;
  if (!tried) {
    tried = true;
    const char* off = getenv("GR_NO_MAILBOX");
    void* mp = nullptr;
    if (!(off && off[0] == '1') && hipHostMalloc(&mp, MAIL_WORDS * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      memset(mp, 0, MAIL_WORDS * sizeof(int32_t));
      page = static_cast<int32_t*>(mp);
    } else {
      (void)hipGetLastError();
    }
  }
  return page;
}

int mailbox_next_stamp() {
  static std::atomic<int> seq{0};
  int s = seq.fetch_add(1) + 1;
  if (s > 0x7ffffff0) {  // two billion calls: start over (a stale stamp of that age cannot be in flight)
    seq.store(1);
    s = 1;
  }
  return s;
}

int mailbox_wait(const volatile int32_t* stamp_word, int stamp, hipStream_t stream, const char* what) {
  // No time limit of its own: as long as the stream is still running the kernel may simply be long (or the device shared);
  // a stream that has drained or failed without the stamp is an error.  (Exactly what hipStreamSynchronize would wait for.)
  for (unsigned long spin = 1;; ++spin) {
    if (__atomic_load_n(stamp_word, __ATOMIC_ACQUIRE) == stamp) return GR_OK;
    __builtin_ia32_pause();  // (eight ranks spin like this on one host: leave the core's other thread its issue slots)
    if ((spin & 0xffff) == 0 && hipStreamQuery(stream) != hipErrorNotReady) {
      GR_HIP(hipStreamSynchronize(stream));
      GR_REQUIRE(__atomic_load_n(stamp_word, __ATOMIC_ACQUIRE) == stamp, "%s: the kernel did not post its result", what);
      return GR_OK;
    }
  }
}

// ------------------------------------------------------------------ scan
constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;  // 2048 elements per block
static_assert(SCAN_TILE == SCAN_SINGLE_ROW, "common.hpp promises single-launch scans up to this row length");

__device__ inline int wave_incl_scan(int v, int lane) {
  (void)lane;
  return wave_incl_scan_add_dpp(v);
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
                                                            int tiles, const int32_t* __restrict__ last_dev,
                                                            int32_t* __restrict__ total_single,
                                                            int32_t* __restrict__ row_max) {
  // total_single / row_max: single-tile rows only (tiles == 1) -- the row's total and its largest element are written
  // here and the other two phases are not launched
  const int row = blockIdx.y;
  const int64_t tile0 = (int64_t)blockIdx.x * SCAN_TILE;
  if (last_dev) {  // only entries 0..*last_dev are wanted (the array is sized for a worst case)
    n = min(n, (int64_t)*last_dev + 1);
    if (tile0 >= n) {
      if (threadIdx.x == 0) partial[(int64_t)row * tiles + blockIdx.x] = 0;
      return;
    }
  }
  const int32_t* src = in + row * row_stride;
  int32_t* dst = out + row * row_stride;
  int v[SCAN_ITEMS];
  int sum = 0, mx = INT32_MIN;
  const int64_t base = tile0 + (int64_t)threadIdx.x * SCAN_ITEMS;
  // whole tiles of 16-byte aligned rows move as int4 (eight 4-byte accesses per thread at a 32-byte lane stride touch every
  // line of the wave's 2 KB eight times)
  static_assert(SCAN_ITEMS == 8, "two int4 per thread");
  const bool vec = tile0 + SCAN_TILE <= n && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (vec) {
    const int4 a = reinterpret_cast<const int4*>(src + base)[0], b = reinterpret_cast<const int4*>(src + base)[1];
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      sum += v[k];
      mx = max(mx, v[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      v[k] = (base + k < n) ? src[base + k] : 0;
      sum += v[k];
      if (base + k < n) mx = max(mx, v[k]);
    }
  }
  if (row_max) {
    __shared__ int s_mx[SCAN_T / WAVE];
    const int wm = wave_max_i32_dpp(mx);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_mx[threadIdx.x / WAVE] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
      int m = s_mx[0];
#pragma unroll
      for (int i = 1; i < SCAN_T / WAVE; ++i) m = max(m, s_mx[i]);
      row_max[row] = m;
    }
  }
  int tot;
  int ex = block_excl_scan(sum, &tot);
  if (vec) {
    int o[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      o[k] = ex;
      ex += v[k];
    }
    reinterpret_cast<int4*>(dst + base)[0] = make_int4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<int4*>(dst + base)[1] = make_int4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      if (base + k < n) dst[base + k] = ex;
      ex += v[k];
    }
  }
  if (threadIdx.x == 0) {
    partial[(int64_t)row * tiles + blockIdx.x] = tot;
    if (total_single) total_single[row] = tot;
  }
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
                                                          int tiles, const int32_t* __restrict__ last_dev) {
  const int row = blockIdx.y;
  if (last_dev) {
    n = min(n, (int64_t)*last_dev + 1);
    if ((int64_t)blockIdx.x * SCAN_TILE >= n) return;
  }
  const int add = partial[(int64_t)row * tiles + blockIdx.x];
  if (add == 0) return;
  int32_t* dst = out + row * row_stride;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  if ((int64_t)(blockIdx.x + 1) * SCAN_TILE <= n && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    int4* d4 = reinterpret_cast<int4*>(dst + base);
    int4 a = d4[0], b = d4[1];
    a.x += add, a.y += add, a.z += add, a.w += add, b.x += add, b.y += add, b.z += add, b.w += add;
    d4[0] = a;
    d4[1] = b;
    return;
  }
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) dst[base + k] += add;
}

// phases 2 + 3 in one launch for rows of up to SCAN_SELF_TILES tiles: every workgroup sums the raw totals of the tiles
// before it itself (a few hundred words out of L2) instead of waiting for a one-workgroup scan of them; the row's last
// tile writes the row total.  One dependent launch less (~5 us) on the launch-bound paths (grid_subsample scans twice).
constexpr int SCAN_SELF_TILES = 1024;
__global__ __launch_bounds__(SCAN_T) void scan_add_self_kernel(int32_t* __restrict__ out, int64_t n, int64_t row_stride,
                                                               const int32_t* __restrict__ partial, int tiles,
                                                               const int32_t* __restrict__ last_dev,
                                                               int32_t* __restrict__ total) {
  const int row = blockIdx.y;
  const int32_t* p = partial + (int64_t)row * tiles;
  int mine = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_T) mine += p[i];
  int add;
  (void)block_excl_scan(mine, &add);
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) total[row] = add + p[blockIdx.x];
  if (last_dev) {
    n = min(n, (int64_t)*last_dev + 1);
    if ((int64_t)blockIdx.x * SCAN_TILE >= n) return;
  }
  if (add == 0) return;
  int32_t* dst = out + row * row_stride;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  if ((int64_t)(blockIdx.x + 1) * SCAN_TILE <= n && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    int4* d4 = reinterpret_cast<int4*>(dst + base);
    int4 a = d4[0], b = d4[1];
    a.x += add, a.y += add, a.z += add, a.w += add, b.x += add, b.y += add, b.z += add, b.w += add;
    d4[0] = a;
    d4[1] = b;
    return;
  }
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) dst[base + k] += add;
}

// ------------------------------------------------------------------ per-cloud bounding boxes
constexpr int BBOX_SLICE = 1024;  // points per step of a block (3072 floats = 256 threads x 12 loads, all in flight together)
constexpr int BBOX_CHUNK = 4 * BBOX_SLICE;  // points per block: its six atomics land on the few cache lines that hold ALL the
                                            // boxes of the call and are served one at a time (~10 ns) -- with 1 024 points per
                                            // block the 75 000 atomics of a 64 x 200 k call WERE the launch (60 us); four steps
                                            // per block: a quarter of them

// Block `blk` reduces one BBOX_CHUNK-point slice of ONE cloud (blk_off[b] = first block of cloud
// b), reading it as a flat, fully coalesced float stream, and issues 6 atomics.
// (Same-address global atomics cost ~11 ns each: one per wave was 200 us for 200 k points.)
// post (optional): ticket = a zeroed counter, mail = the host's mailbox page -- the LAST workgroup to finish copies the 6 nb
// words to mail[0 .. 6 nb) and stamps mail[6 nb] (common.hpp): the host gets the boxes without a copy in the stream.
struct BboxPost {
  int32_t* ticket;
  int32_t* mail;
  int stamp;
};
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ pts,
                                                   const int32_t* __restrict__ off,
                                                   const int32_t* __restrict__ blk_off, int nb,
                                                   uint32_t* __restrict__ bbox, BboxPost post) {
  __shared__ uint32_t red[6][256 / WAVE];
  __shared__ int s_last;
  __shared__ int s_where[3];
  __shared__ int32_t s_tab[256];
  // which cloud: a search through blk_off -- seven DEPENDENT loads in front of the data loads of every workgroup (3 - 4 us of a
  // 5 us workgroup).  Small batches: the table comes into LDS with one load and is searched there.
  if (nb + 1 <= 256) {
    if ((int)threadIdx.x <= nb) s_tab[threadIdx.x] = blk_off[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int bb = nb + 1 <= 256 ? find_batch(s_tab, nb, (int)blockIdx.x) : find_batch(blk_off, nb, (int)blockIdx.x);
    s_where[0] = bb;
    s_where[1] = off[bb] + ((int)blockIdx.x - (nb + 1 <= 256 ? s_tab[bb] : blk_off[bb])) * BBOX_CHUNK;
    s_where[2] = off[bb + 1];
  }
  __syncthreads();
  const int b0 = s_where[0];
  const int p_first = s_where[1];
  const int p_end = min(s_where[2], p_first + BBOX_CHUNK);
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  for (int ps = p_first; ps < p_end; ps += BBOX_SLICE) {  // (block-uniform trip count)
  const int64_t f0 = (int64_t)ps * 3, f1 = (int64_t)min(p_end, ps + BBOX_SLICE) * 3;
  // the axis of element f is f % 3 and the stride is 256 = 1 (mod 3): a thread's elements cycle through the axes, so
  // three consecutive loads (issued together) feed lo/hi[ax], [ax+1], [ax+2] -- no modulo, no dependent load chain
  const int ax0 = (int)((f0 + threadIdx.x) % 3);
  const float* src = pts + f0;
  const int count = (int)(f1 - f0);
  uint32_t l3[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, h3[3] = {0u, 0u, 0u};  // indexed by (axis - ax0) mod 3
  // all twelve loads of the thread first, on clamped indices and without branches: ONE memory round trip per step
  // (four batches of three, each behind the previous one's use, were four: 94 us for 64 x 200 k points)
  constexpr int PER = 3 * BBOX_SLICE / 256;
  static_assert(PER % 3 == 0, "a thread's elements must cycle through the axes");
  static_assert(PER == 12, "the aligned path reads a thread's share as three float4");
  const bool wide = count == 3 * BBOX_SLICE && (reinterpret_cast<uintptr_t>(src) & 15u) == 0;  // block-uniform
  if (wide) {
    // a full, 16-byte aligned slice: thread t owns four whole points (floats 12 t .. 12 t + 11) as three 16-byte loads
    const float4* s4 = reinterpret_cast<const float4*>(src) + 3 * threadIdx.x;
    const float4 a = s4[0], bq = s4[1], cq = s4[2];  // x y z x | y z x y | z x y z
    const float ex[4] = {a.x, a.w, bq.z, cq.y}, ey[4] = {a.y, bq.x, bq.w, cq.z}, ez[4] = {a.z, bq.y, cq.x, cq.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t vx = f2ord(ex[u]), vy = f2ord(ey[u]), vz = f2ord(ez[u]);
      lo[0] = min(lo[0], vx), hi[0] = max(hi[0], vx);
      lo[1] = min(lo[1], vy), hi[1] = max(hi[1], vy);
      lo[2] = min(lo[2], vz), hi[2] = max(hi[2], vz);
    }
  } else {
    float raw[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) raw[u] = src[min((int)threadIdx.x + u * 256, max(count - 1, 0))];
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if ((int)threadIdx.x + u * 256 < count) {
        const uint32_t v = f2ord(raw[u]);
        l3[u % 3] = min(l3[u % 3], v);
        h3[u % 3] = max(h3[u % 3], v);
      }
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // slot k holds axis (ax0 + k) % 3
      const int ax = ax0 + k >= 3 ? ax0 + k - 3 : ax0 + k;
      lo[0] = ax == 0 ? min(lo[0], l3[k]) : lo[0];
      lo[1] = ax == 1 ? min(lo[1], l3[k]) : lo[1];
      lo[2] = ax == 2 ? min(lo[2], l3[k]) : lo[2];
      hi[0] = ax == 0 ? max(hi[0], h3[k]) : hi[0];
      hi[1] = ax == 1 ? max(hi[1], h3[k]) : hi[1];
      hi[2] = ax == 2 ? max(hi[2], h3[k]) : hi[2];
    }
  }
  }
  // wave-wide on the DPP network (the ordered words, biased into int range): six VALU steps per value instead of six
  // ds_bpermute round trips
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = (uint32_t)wave_min_i32_dpp((int)(lo[k] ^ 0x80000000u)) ^ 0x80000000u;
    hi[k] = (uint32_t)wave_max_i32_dpp((int)(hi[k] ^ 0x80000000u)) ^ 0x80000000u;
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      red[k][w] = lo[k];
      red[3 + k][w] = hi[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    uint32_t v = red[threadIdx.x][0];
#pragma unroll
    for (int i = 1; i < 256 / WAVE; ++i)
      v = threadIdx.x < 3 ? min(v, red[threadIdx.x][i]) : max(v, red[threadIdx.x][i]);
    if (threadIdx.x < 3) atomicMin(&bbox[b0 * 6 + threadIdx.x], v);
    else atomicMax(&bbox[b0 * 6 + threadIdx.x], v);
  }
  if (post.mail == nullptr) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(post.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  for (int i = threadIdx.x; i < 6 * nb; i += 256)
    post.mail[i] = (int32_t)__hip_atomic_load(&bbox[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) mail_post(post.mail + 6 * nb, post.stamp);
}

__global__ void bbox_init_kernel(uint32_t* __restrict__ bbox, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb * 6) bbox[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
}


void bbox_block_offsets(const int32_t* h_off, int32_t* blk, int nb) {
  blk[0] = 0;
  for (int b = 0; b < nb; ++b) blk[b + 1] = blk[b] + (h_off[b + 1] - h_off[b] + BBOX_CHUNK - 1) / BBOX_CHUNK;
}

int compute_bbox(const float* pts, const int32_t* h_off, int32_t* blk, const int32_t* off_dev, int nb,
                 uint32_t* bbox_dev, int32_t* blk_off_dev, hipStream_t stream, bool blk_off_on_device, bool init_bbox,
                 int32_t* zeroed_ticket, int32_t* mail, int stamp) {
  if (nb <= 0) return GR_OK;
  if (!blk_off_on_device) {
    bbox_block_offsets(h_off, blk, nb);
    GR_HIP(hipMemcpyAsync(blk_off_dev, blk, sizeof(int32_t) * (nb + 1), hipMemcpyHostToDevice, stream));
  }
  if (init_bbox) hipLaunchKernelGGL(bbox_init_kernel, dim3((nb * 6 + 255) / 256), dim3(256), 0, stream, bbox_dev, nb);
  if (blk[nb] > 0)
    hipLaunchKernelGGL(bbox_kernel, dim3(blk[nb]), dim3(256), 0, stream, pts, off_dev, blk_off_dev, nb, bbox_dev,
                       BboxPost{zeroed_ticket, zeroed_ticket ? mail : nullptr, stamp});
  GR_LAUNCH_CHECK();
  return GR_OK;
}

size_t scan_ws_ints(int64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1; }

int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int rows, int64_t row_stride,
                       int32_t* scan_ws, int32_t* total, hipStream_t stream, const int32_t* last_dev, int32_t* row_max) {
  if (n <= 0 || rows <= 0) return GR_OK;
  const int tiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  GR_REQUIRE(row_max == nullptr || tiles == 1, "exclusive_scan_i32: row maxima need rows of at most %d elements", SCAN_TILE);
  if (tiles == 1) {  // one workgroup per row does everything: one launch instead of two
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1, rows), dim3(SCAN_T), 0, stream, in, out, n, row_stride, scan_ws, 1,
                       last_dev, total, row_max);
    GR_LAUNCH_CHECK();
    return GR_OK;
  }
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(tiles, rows), dim3(SCAN_T), 0, stream, in, out, n,
                     row_stride, scan_ws, tiles, last_dev, (int32_t*)nullptr, (int32_t*)nullptr);
  if (tiles <= SCAN_SELF_TILES) {
    hipLaunchKernelGGL(scan_add_self_kernel, dim3(tiles, rows), dim3(SCAN_T), 0, stream, out, n, row_stride, scan_ws, tiles,
                       last_dev, total);
    GR_LAUNCH_CHECK();
    return GR_OK;
  }
  hipLaunchKernelGGL(scan_partials_kernel, dim3(rows), dim3(SCAN_T), 0, stream, scan_ws, tiles,
                     total);
  if (tiles > 1)
    hipLaunchKernelGGL(scan_add_kernel, dim3(tiles, rows), dim3(SCAN_T), 0, stream, out, n,
                       row_stride, scan_ws, tiles, last_dev);
  GR_LAUNCH_CHECK();
  return GR_OK;
}


// ---------------------------------------------------------------- LDS atomic ordering probe
namespace {
// Hardware property probe.  gfx950's LDS resolves the lanes of ONE ds_add_rtn_u32 that hit the same address in ascending
// lane order (the returned pre-add values grow with the lane id).  That is not an architectural promise, so it is
// MEASURED once per process and device, in the shape the production kernels use it: 16 waves per workgroup, every wave on
// its own 512-counter row (the depth sort's layout), 256 workgroups in flight so that all waves of a CU contend for the
// LDS at once, 96 address patterns (strided, hashed, with lanes masked off).  If any group of equal-address lanes comes
// back out of lane order the sorts use explicit ballot ranking.  The probe is backed by a check of the real output: the
// rasterizer verifies the depth order and the per-tile lists of its first frames (rasterizer.hip) and demotes the device
// (lds_order_demote) if they are not sorted.
constexpr int PROBE_WAVES = 16, PROBE_ROW = 512;

__global__ __launch_bounds__(PROBE_WAVES* WAVE) void lds_atomic_order_probe_kernel(int* __restrict__ bad) {
  __shared__ unsigned int cell[PROBE_WAVES][PROBE_ROW];
  __shared__ unsigned int got[PROBE_WAVES][WAVE];
  __shared__ unsigned int adr[PROBE_WAVES][WAVE];
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int nbad = 0;
  for (int pat = 0; pat < 96; ++pat) {
    for (int i = lane; i < PROBE_ROW; i += WAVE) cell[w][i] = 0u;
    __syncthreads();
    unsigned int a;
    if (pat < 64) a = (unsigned int)(lane % (pat + 1)) * 7u;                 // 1 .. 64 distinct addresses, strided
    else a = ((unsigned int)((lane + 64 * w) * 2654435761u + (pat + blockIdx.x) * 40503u) >> 7) % (unsigned int)(3 + (pat - 64) * 15);
    const bool take = pat < 80 || ((lane * 7 + pat + w) % 5) != 0;            // some patterns run with lanes masked off
    unsigned int r = 0xffffffffu;
    if (take) r = atomicAdd(&cell[w][a], 1u);
    got[w][lane] = r;
    adr[w][lane] = a;
    __syncthreads();
    if (take) {
      unsigned int want = 0;  // lanes below me on the same address that took part
      for (int l = 0; l < lane; ++l) want += (adr[w][l] == a && got[w][l] != 0xffffffffu) ? 1u : 0u;
      if (want != r) ++nbad;
    }
    __syncthreads();
  }
  if (nbad) atomicAdd(bad, nbad);
}

}  // namespace

// 1 = lane-ordered LDS atomics verified on this device, 0 = not (ballot ranking is used), -1 = not probed yet
static std::atomic<int> g_lds_order[64];
static std::once_flag g_lds_once;
static void lds_order_init() {
  for (auto& v : g_lds_order) v.store(-1);
}

// 1 = explicit ballot ranking whatever the probe says (GR_RASTER_BALLOT_RANKING=1 or gr_raster_ballot_ranking(1))
static std::atomic<int>& lds_force_ballot() {
  static std::atomic<int> v{[] {
    const char* force = getenv("GR_RASTER_BALLOT_RANKING");
    return (force && force[0] == '1') ? 1 : 0;
  }()};
  return v;
}

static int lds_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  return (dev < 0 || dev >= 64) ? 63 : dev;
}

int lds_atomics_lane_ordered(hipStream_t stream, bool* ordered) {
  std::call_once(g_lds_once, lds_order_init);
  const int dev = lds_device_slot();
  GR_REQUIRE(dev >= 0, "hipGetDevice failed");
  if (lds_force_ballot().load() == 1) {
    *ordered = false;
    return GR_OK;
  }
  if (g_lds_order[dev].load() < 0) {
    int* d_bad = nullptr;
    int h_bad = 1;
    GR_HIP(hipMallocAsync(reinterpret_cast<void**>(&d_bad), sizeof(int), stream));
    GR_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), stream));
    hipLaunchKernelGGL(lds_atomic_order_probe_kernel, dim3(256), dim3(PROBE_WAVES * WAVE), 0, stream, d_bad);
    GR_HIP(hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    GR_HIP(hipFreeAsync(d_bad, stream));
    int expect = -1;
    g_lds_order[dev].compare_exchange_strong(expect, h_bad == 0 ? 1 : 0);  // a concurrent demotion is not overwritten
  }
  *ordered = g_lds_order[dev].load() == 1;
  return GR_OK;
}

int lds_atomics_lane_ordered_state() {
  std::call_once(g_lds_once, lds_order_init);
  const int dev = lds_device_slot();
  if (dev >= 0 && lds_force_ballot().load() == 1) return 0;
  return dev < 0 ? -1 : g_lds_order[dev].load();
}

int lds_ballot_ranking_force(int on) {
  const int old = lds_force_ballot().load();
  if (on == 0 || on == 1) lds_force_ballot().store(on);
  return old;
}

void lds_order_demote() {
  std::call_once(g_lds_once, lds_order_init);
  const int dev = lds_device_slot();
  if (dev >= 0) g_lds_order[dev].store(0);
}

}  // namespace gr

extern "C" const char* gr_last_error(void) { return gr::g_err; }
extern "C" int gr_version(void) { return 1000; }

extern "C" void gr_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(gr::g_tmutex);
  gr::g_timing_on = on != 0;
}

extern "C" void gr_timing_reset(void) {
  std::lock_guard<std::mutex> lk(gr::g_tmutex);
  for (auto& r : gr::g_trecs) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  gr::g_trecs.clear();
}

extern "C" int gr_timing_read(const char* name, double* total_ms, int64_t* calls) {
  std::lock_guard<std::mutex> lk(gr::g_tmutex);
  double tot = 0.0;
  int64_t n = 0;
  for (auto& r : gr::g_trecs) {
    if (r.name != name) continue;
    if (hipEventSynchronize(r.stop) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      tot += ms;
      ++n;
    }
  }
  if (total_ms) *total_ms = tot;
  if (calls) *calls = n;
  return GR_OK;
}
