// Shared host/device helpers for libgaussreg_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/gaussreg_hip.h"

namespace gr {

constexpr int WAVE = 64;
// gfx950 only: several kernels of this library allocate 64 - 93 KB of static LDS per workgroup (the in-LDS bucket sorts of
// depth_sort.hip, the slab chains of hash_order_device.hip, the tile scatter), which needs the 160 KB a CDNA4 CU has;
// on any other --offload-arch this fails here with a reason instead of a bare "local memory limit exceeded".
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgaussreg_hip is written for gfx950 (MI355X, 160 KB of LDS per CU): build with --offload-arch=gfx950"
#endif

void set_error(const char* fmt, ...);

#define GR_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t _e = (call);                                                                \
    if (_e != hipSuccess) {                                                                \
      gr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return GR_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

#define GR_REQUIRE(cond, ...)       \
  do {                              \
    if (!(cond)) {                  \
      gr::set_error(__VA_ARGS__);   \
      return GR_ERR_INVALID;        \
    }                               \
  } while (0)

#define GR_LAUNCH_CHECK() GR_HIP(hipGetLastError())

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace (256-B aligned carves).
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* r = reinterpret_cast<T*>(base ? base + off : nullptr);
    off += sizeof(T) * count;
    return r;
  }
  size_t used() const { return align_up(off, 256); }
};

// Order-preserving float <-> uint map (for atomicMin/Max on floats).
__host__ __device__ inline uint32_t f2ord(float f) {
  uint32_t u;
#ifdef __HIP_DEVICE_COMPILE__
  u = __float_as_uint(f);
#else
  memcpy(&u, &f, 4);
#endif
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  float f;
#ifdef __HIP_DEVICE_COMPILE__
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}

// ---- exclusive scan of int32 arrays (rows of equal length), 3 small launches ----------------
// scan_ws needs scan_ws_ints(n) int32 per row.
size_t scan_ws_ints(int64_t n);
// in/out may alias.  out[i] = sum_{j<i} in[j]; total[r] (optional, device) = row sum.
// last_dev (optional, device): only out[0..*last_dev] is needed -- tiles past it are skipped (arrays sized for a
// worst case whose real extent is only known on the device).
// Rows of at most SCAN_SINGLE_ROW elements take one launch; for those, row_max[r] (optional, device) = largest element.
constexpr int SCAN_SINGLE_ROW = 2048;
int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int rows, int64_t row_stride,
                       int32_t* scan_ws, int32_t* total, hipStream_t stream, const int32_t* last_dev = nullptr,
                       int32_t* row_max = nullptr);

// Pinned host scratch, one buffer per (host thread, slot), grown on demand and kept for the life of the process.  A copy
// to or from pageable memory is staged and synchronised by the runtime; through these buffers the small uploads and
// read-backs of the API calls are asynchronous for real.  Contract: the caller synchronises the stream before it returns
// (slots 0-3 and 5-8: every entry point that uses them does) or guards the slot with an event it waits on before the next use
// (slot 4, hash_order_device), so the next call on the thread finds the buffer free.
void* pinned_scratch(int slot, size_t bytes);

// Host-mapped, coherent pinned words (one 4 KB page per host thread) that a kernel writes and the host polls: a small
// read-back without a copy in the stream and without a stream synchronise.  The kernel stores its payload, then the
// call's stamp with a system-scope release store (mail_post); the host spins on the stamp (hipStreamQuery every
// 65 536 polls so a failed stream is noticed).  mailbox() returns nullptr when the page cannot be mapped or
// GR_NO_MAILBOX=1 is set -- callers then copy and synchronise as before.
// Every user owns a disjoint region of the page (a word that is payload for one operator is never the stamp word of
// another) and clears its stamp word on the host before the launch that will post it (mailbox_arm).
constexpr int MAIL_WORDS = 1024;
constexpr int MAIL_GRID_BOXES = 0;     // grid_subsample: 6 B box words + stamp, B <= 80
constexpr int MAIL_GRID_COUNTS = 512;  // grid_subsample: B + 1 counts, the bucket-overflow flag, the stamp, B <= 256
constexpr int MAIL_RADIUS = 1008;      // radius search: 5 header words + stamp
volatile int32_t* mailbox();
inline void mailbox_arm(volatile int32_t* stamp_word) { __atomic_store_n(stamp_word, 0, __ATOMIC_RELEASE); }
int mailbox_next_stamp();  // per process, never 0
int mailbox_wait(const volatile int32_t* stamp_word, int stamp, hipStream_t stream, const char* what);
__device__ __forceinline__ void mail_post(int32_t* stamp_word, int stamp) {
  __hip_atomic_store(stamp_word, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Per-cloud bounding boxes of stacked points: bbox_dev[b*6 + {0,1,2}] = min xyz, +{3,4,5} = max xyz,
// stored as order-preserving uints (decode with ord2f).  Empty clouds keep (0xffffffff, 0).
// h_off: host copy of the nb+1 point offsets; h_blk: nb+1 ints of HOST scratch that must stay
// alive until the stream is synchronised; blk_off_dev: nb+1 ints of device scratch.
// blk_off_on_device: the caller has filled h_blk with bbox_block_offsets() and copied it to blk_off_dev itself (to
// merge several small host-to-device copies into one).
void bbox_block_offsets(const int32_t* h_off, int32_t* h_blk, int nb);
int compute_bbox(const float* pts, const int32_t* h_off, int32_t* h_blk, const int32_t* off_dev, int nb,
                 uint32_t* bbox_dev, int32_t* blk_off_dev, hipStream_t stream, bool blk_off_on_device = false,
                 bool init_bbox = true,  // init_bbox = false: the caller has set bbox_dev to (0xffffffff x3, 0 x3) per cloud
                 // zeroed_ticket + mail (the mailbox page, >= 6 nb + 1 words) + stamp: the last workgroup posts the boxes to
                 // mail[0 .. 6 nb) and the stamp to mail[6 nb] -- nothing is posted when there are no points at all
                 int32_t* zeroed_ticket = nullptr, int32_t* mail = nullptr, int stamp = 0);

// Stable LSD radix sort of (u64 key, i32 value) pairs on bits [begin_bit, end_bit) (sort.hip: this library's own kernels).
size_t sort_pairs_temp_bytes(int64_t n);
int sort_pairs_u64_i32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                       const int32_t* vals_in, int32_t* vals_out, int64_t n, int begin_bit,
                       int end_bit, hipStream_t stream);
// value stream generated on the fly: value(i) = i mod period
int sort_pairs_u64_iota(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, int64_t period,
                        int32_t* vals_out, int64_t n, int begin_bit, int end_bit, hipStream_t stream);

// Do the lanes of ONE ds_add_rtn_u32 that hit the same LDS address get their pre-add values in ascending lane order on
// this device?  Probed once per process and device (common.hip); GR_RASTER_BALLOT_RANKING=1 forces "no".
int lds_atomics_lane_ordered(hipStream_t stream, bool* ordered);
int lds_atomics_lane_ordered_state();  // 1 yes, 0 no, -1 not probed yet
int lds_ballot_ranking_force(int on);   // 1 = ballot ranking whatever the probe says, 0 = as probed; returns the previous setting
void lds_order_demote();                // a sort that relied on the property came out unsorted: use ballot ranking from now on

// ---- the rasterizer's packed tile rectangles (26 bits: x:7 | y:7 | w:6 | h:6; shared by rasterizer.hip and depth_sort.hip)
constexpr int RASTER_TILE = 16;
constexpr uint32_t RECT_MARKER26 = 127u | (127u << 7);  // w = h = 0: "did not fit, rebuild from the record"
#ifdef __HIPCC__
__device__ __forceinline__ void get_rect(float px, float py, int radius, int gx, int gy, int* rmin, int* rmax) {
  const float r = (float)radius;
  rmin[0] = min(gx, max(0, (int)((px - r) / (float)RASTER_TILE)));
  rmin[1] = min(gy, max(0, (int)((py - r) / (float)RASTER_TILE)));
  rmax[0] = min(gx, max(0, (int)((px + r + (float)(RASTER_TILE - 1)) / (float)RASTER_TILE)));
  rmax[1] = min(gy, max(0, (int)((py + r + (float)(RASTER_TILE - 1)) / (float)RASTER_TILE)));
}
// rec: the per-(view, Gaussian) records ([.][4] float4: [0] = {px, py, ..}, [3].x = radius as int bits)
__device__ __forceinline__ bool rect_decode(uint32_t r, int id, int64_t vbase, const float4* __restrict__ rec, int gx,
                                            int gy, int& x0, int& y0, int& w, int& h) {
  x0 = (int)(r & 127u);
  y0 = (int)((r >> 7) & 127u);
  w = (int)((r >> 14) & 63u);
  h = (int)((r >> 20) & 63u);
  if (r == RECT_MARKER26) {  // wide rectangle: rebuild the reference square from the record
    const float4 r0 = rec[4 * (vbase + id)];
    int rmin[2], rmax[2];
    get_rect(r0.x, r0.y, __float_as_int(rec[4 * (vbase + id) + 3].x), gx, gy, rmin, rmax);
    x0 = rmin[0];
    y0 = rmin[1];
    w = rmax[0] - rmin[0];
    h = rmax[1] - rmin[1];
  }
  return w * h != 0;
}
#endif

// Segmented stable LSD radix sort of the rasterizer's (view, Gaussian) depth keys (depth_sort.hip).
// (bucket path only) the sort also adds up the tile instances of every `chunk` consecutive depth-ordered Gaussians of a
// view -- what the binning needs first -- into chunk_total[v * nchunk + c], which it clears itself
struct DepthSortTotals {
  int32_t* chunk_total;
  int chunk, nchunk;
  const float4* rec;  // records, for rectangles that do not fit the packing
  int gx, gy;
};
// (bucket path only) ragged segments instead of V equal strides: segment v is [seg_off[v], seg_off[v + 1]) of every array
// (P = the longest one), its key range is known to the caller (range_in[2 v] = smallest field, [2 v + 1] = the shift s with
// (largest - smallest) >> s < 512), ids leave as positions in the whole array, and the 26-bit payload can leave widened
// to 64-bit (segment << key64_shift | payload) words instead of rect_out -- grid_subsample's (cloud, voxel) sort.
struct DepthSortSegments {
  const int32_t* seg_off;
  const uint32_t* range_in;
  uint64_t* key64_out;
  int key64_shift;
  int32_t* big_cnt;   // (set by depth_sort_views: the list the small-bucket launch leaves for the large-bucket one)
  int32_t* big_list;
  const uint64_t* gather64;  // non-null: key64_out[position] = gather64[the entry's index in the whole array] instead
  int bins_used = 0;         // > 0: no segment's top digit reaches this value (small segments sorted on fewer than nine bits:
                             // the scan and the bucket launch skip the digits above)
  int xcd_segments = 0;      // (set by depth_sort_views) > 0: the small-bucket launch is a 1-D grid that deals all buckets of a
                             // segment to ONE XCD (workgroup b runs on XCD b % 8); the value is the number of segments
};
size_t depth_sort_table_bytes(int64_t P, int V);
int depth_sort_views(const uint32_t* field, const uint32_t* rect_raw, uint64_t* keys_a, uint64_t* keys_b, int32_t* ids_out,
                     uint32_t* rect_out, int32_t* nvalid_out, int64_t P, int V, int key_bits, void* table, size_t table_bytes,
                     hipStream_t stream, const int2* key_mm = nullptr, int nb_mm = 0, int32_t* overflow_flag = nullptr,
                     int overflow_value = 0, const DepthSortTotals* chunk_totals = nullptr,
                     const DepthSortSegments* segments = nullptr);
// key_mm != null: the four-launch path for a few views per call (top-digit pass + in-LDS bucket sort): key_mm = [V][nb_mm]
// {smallest, largest} non-zero field of a block of Gaussians (0x7fffffff / 0 for a block without one); a bucket that does
// not fit stores overflow_value into *overflow_flag (a negative value; a positive one is OR-ed in) and the order is then NOT
// valid -- repeat with key_mm = null.
bool depth_sort_msd_possible(int64_t P, int V, int key_bits);

// Optional per-kernel HIP-event timing (off by default; bench.py turns it on to measure the
// dominant kernel's average launch duration on the stream it is launched on).
struct KernelTimer {
  KernelTimer(const char* name, hipStream_t stream);
  ~KernelTimer();
  KernelTimer(const KernelTimer&) = delete;
  int slot_;
  hipStream_t stream_;
};

// Host: iteration order of std::unordered_map<size_t,...> after inserting `keys` (distinct) in order;
// perm_out[j] = base + index of the j-th iterated key (hash_order.hip).
void unordered_map_order(const uint64_t* keys, int64_t n, int32_t base, int32_t* perm_out);
// The same order for many clouds at once, evaluated on the device (hash_order_device.hip): keys on the device, clouds
// contiguous (h_begins: batch + 1 host offsets); perm_out[begin_c + j] = global index of the j-th iterated key of cloud c.
size_t hash_order_device_bytes(int64_t n, int64_t batch);
// rows_out (optional): rows_out[begin_c + j] = rows_in[row_of[that key's index]], 3 floats per row -- the caller's gather
// by the permutation done by the launch that knows the final positions; perm_out may then be null.
int hash_order_device(const uint64_t* keys, const int64_t* h_begins, int64_t batch, int32_t* perm_out, void* ws,
                      size_t ws_bytes, hipStream_t stream, const float* rows_in = nullptr, const int32_t* row_of = nullptr,
                      float* rows_out = nullptr);

// lower_bound over a small ascending int32 offsets table: largest b with off[b] <= i (b < nb)
// ---- wave64 scan / reduction on the DPP shift network (row_shr 1/2/4/8 inside 16-lane rows, row_bcast 15/31 across
// rows): six dependent VALU ops instead of six ds_bpermute round trips.  `old` = identity for lanes without a source.
__device__ __forceinline__ int wave_incl_scan_add_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// wave-wide sum of a double on the same network (two 32-bit DPP moves + one v_add_f64 per step; lanes without a source
// add +0.0), fixed order; the result is valid in every lane.  A __shfl_xor butterfly on doubles is 12 ds_bpermute trips.
__device__ __forceinline__ double wave_sum_f64_dpp(double x) {
#define GR_F64_STEP(CTRL, ROWMASK)                                                                         \
  {                                                                                                        \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xf, false);           \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xf, false);           \
    x += __hiloint2double(hi, lo);                                                                         \
  }
  GR_F64_STEP(0x111, 0xf) GR_F64_STEP(0x112, 0xf) GR_F64_STEP(0x114, 0xf) GR_F64_STEP(0x118, 0xf)
  GR_F64_STEP(0x142, 0xa) GR_F64_STEP(0x143, 0xc)
#undef GR_F64_STEP
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}

// wave-wide min / max / sum of an int on the same network; the result is valid in every lane (read back from lane 63)
#define GR_DPP_REDUCE(NAME, OP, IDENT)                                                         \
  __device__ __forceinline__ int NAME(int v) {                                                 \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x111, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x112, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x114, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x118, 0xf, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x142, 0xa, 0xf, false));                  \
    v = OP(v, __builtin_amdgcn_update_dpp(IDENT, v, 0x143, 0xc, 0xf, false));                  \
    return __builtin_amdgcn_readlane(v, 63);                                                   \
  }
#define GR_OP_MIN(a, b) min(a, b)
#define GR_OP_MAX(a, b) max(a, b)
#define GR_OP_ADD(a, b) ((a) + (b))
GR_DPP_REDUCE(wave_min_i32_dpp, GR_OP_MIN, 0x7fffffff)
GR_DPP_REDUCE(wave_max_i32_dpp, GR_OP_MAX, (int)0x80000000)
GR_DPP_REDUCE(wave_sum_i32_dpp, GR_OP_ADD, 0)
#undef GR_DPP_REDUCE

__device__ inline int find_batch(const int32_t* __restrict__ off, int nb, int32_t i) {
  int lo = 0, hi = nb;  // off has nb+1 entries
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

}  // namespace gr
