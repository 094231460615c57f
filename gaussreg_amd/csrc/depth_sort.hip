// Segmented stable LSD radix sort for the rasterizer's depth ordering, hand-written for gfx950 (replaces the rocPRIM call
// the round-1 pipeline made for this step).
//
// Problem shape: V views x P Gaussians, per (view, Gaussian) a 32-bit depth field (0 = culled; `key_bits` significant
// bits) and a packed tile rectangle.  Wanted per view: the visible Gaussians ordered by depth field, ties by Gaussian id
// (= stable), as ids + rectangles, plus the visible count.  The sort moves ONE 8-byte word per entry.  The first pass
// consumes the low 9 field bits, so from then on the word only has to hold the REMAINING field bits:
//   packed   (remaining field bits + id bits + 26 rectangle bits <= 64, e.g. 18 + 20 + 26 for 2^20 Gaussians): the first
//            pass picks the rectangle up with a coalesced read (ids are still sequential there) and the word is
//            [field >> 9 | id | rectangle]; the last pass just unpacks it;
//   gather   (otherwise): the word is (field << 32 | id) and the last pass looks the rectangle up by id -- a random 4-byte
//            gather that costs as much as a whole pass (measured: last pass 0.12 -> 0.24 ms at 32 views x 1 M).
// What is specific here and a general device sort cannot exploit:
//   * segments (views) are contiguous and equal-stride, so no view bits are sorted: 27 depth bits = THREE 9-bit passes
//     (the rocPRIM path needed four 8-bit passes over 27 + 5 view bits);
//   * the first pass reads the bare 4-byte fields, drops the culled entries (40 % of the (view, Gaussian) pairs of the
//     benchmark scene) while it scatters -- compaction costs nothing extra -- and generates the id half of the word;
//   * the last pass writes the id and the 4-byte rectangle it looks up, not the 8-byte word;
//   * ranking inside a 2048-key chunk uses one LDS atomic per key: wave w owns a contiguous quarter of the chunk and a
//     private digit-counter row; keys are taken 64 at a time in position order, and equal-digit lanes of one
//     ds_add_rtn_u32 are served in lane order (probed on the device, common.hip) -- so the returned counts ARE the stable
//     ranks.  Without that property the lanes rank themselves with one ballot per digit bit.
//   * words are staged in LDS at their chunk-local sorted position and leave as contiguous runs per digit, never as
//     scattered 8-byte stores (one stream of 8-byte words: a separate 4-byte id stream halved the run length in bytes);
//   * with >= 8 views all chunks of a view run on one XCD (block b sits on XCD b % 8): the 512 write frontiers of a view
//     stay in that XCD's L2.
// Per pass: count (per-chunk digit histogram) -> scan (prefix over the chunks of a view, digit bases) -> scatter.
//
// Few views per call (one camera: the launch chain IS the frame time, nine launches of 5-9 us): ONE such pass on the TOP
// bits, then one launch that finishes every bucket inside LDS ("MSD" below, four launches).  The top digit is taken
// relative to the view's own key range -- bucket = (key - kmin) >> s with s = bits(kmax - kmin) - 9, from the per-block
// minima / maxima the preprocess kernel leaves -- so the 512 buckets cover exactly the depths that occur; what is left of
// a key is s <= 18 bits.  A bucket (a contiguous run of <= MSD_CAP words, already in id order) becomes 32-bit items
// [rest (s bits) | position in the bucket (14 bits)] -- all distinct, so plain integer order of the items IS (key, id)
// order -- sorted by one or two stable 9-bit passes between two LDS buffers, then the words are fetched in that order.
// A bucket above MSD_CAP raises a flag: the caller repeats the frame with the three-pass sort.
// The same launch can add up the tile instances of every chunk of sorted positions (DepthSortTotals: what the binning needs
// first, so that no count launch follows the sort).
//
// A second user, grid_subsample.hip (DepthSortSegments): ragged segments (clouds) instead of equal strides, key range known
// to the caller, the 26-bit payload = the voxel key, ids leaving as global point indices -- the (cloud, voxel key, index)
// sort of a batch of clouds in two trips through memory.  Its buckets are small (a few hundred points) and many (512 per
// cloud): they run on 128-thread workgroups, which list the few larger ones for a fixed grid of the big ones.
#include <type_traits>

#include "common.hpp"

namespace gr {
namespace {

constexpr int DS_T = 256;
constexpr int DS_NW = DS_T / WAVE;
constexpr int DS_CHUNK = 2048;
constexpr int DS_BITS = 9;
constexpr int DS_BINS = 1 << DS_BITS;
constexpr int DS_STEPS = DS_CHUNK / DS_T;  // keys per thread

__device__ __forceinline__ bool ds_block(int V, int nchunk, int& v, int& c) {
  const int b = blockIdx.x;
  if (V >= 8) {
    const int k = b >> 3;
    v = (b & 7) + 8 * (k / nchunk);
    c = k % nchunk;
    return v < V;
  }
  v = b / nchunk;
  c = b % nchunk;
  return true;
}
inline int ds_grid(int V, int nchunk) { return V >= 8 ? 8 * ((V + 7) / 8) * nchunk : V * nchunk; }

constexpr int MSD_T = 512;         // threads of a bucket workgroup
constexpr int MSD_IDX_BITS = 14;
constexpr int MSD_CAP = 7936;      // items per bucket: 2 x 31 KB of items + 16 KB of counters -> two workgroups per CU
static_assert(MSD_CAP <= (1 << MSD_IDX_BITS), "bucket positions must fit the item's index field");

// FIRST: the source is the raw key array (all P entries of the view, culled ones have a zero depth field)
// MSD (with FIRST): the digit is the top of the key relative to the view's range; every block derives the range from the
// per-block minima / maxima of the preprocess kernel (key_mm: [V][nb_mm] {min, max}, 0x7fffffff / 0 for an empty block)
// and the first one of a view leaves {kmin, s} in range[v] for the launches behind this one.
template <bool FIRST, bool MSD>
__global__ __launch_bounds__(DS_T) void ds_count_kernel(int P, int V, int nchunk, const int32_t* __restrict__ nvalid,
                                                        const uint32_t* __restrict__ field, const uint64_t* __restrict__ keys,
                                                        int word_shift, uint32_t dmask, uint16_t* __restrict__ hist,
                                                        const int2* __restrict__ key_mm, int nb_mm,
                                                        uint32_t* __restrict__ range, DepthSortSegments sg) {
  __shared__ unsigned int s_h[DS_BINS];
  __shared__ int s_mm[2][DS_NW];
  int v, c;
  if (!ds_block(V, nchunk, v, c)) return;
  for (int d = threadIdx.x; d < DS_BINS; d += DS_T) s_h[d] = 0u;
  uint32_t kmin = 0u;
  int sh = 0;
  if (MSD && key_mm == nullptr) {  // the caller knows the key range of every segment (sg.range_in, also read by the launches behind)
    kmin = sg.range_in[2 * v];
    sh = (int)sg.range_in[2 * v + 1];
  } else if (MSD) {
    int mn = 0x7fffffff, mx = 0;
    const int2* row = key_mm + (int64_t)v * nb_mm;
#pragma unroll 4
    for (int i = threadIdx.x; i < nb_mm; i += DS_T) {
      const int2 m = row[i];
      mn = min(mn, m.x);
      mx = max(mx, m.y);
    }
    mn = wave_min_i32_dpp(mn);
    mx = wave_max_i32_dpp(mx);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_mm[0][threadIdx.x / WAVE] = mn, s_mm[1][threadIdx.x / WAVE] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DS_NW; ++w) mn = min(mn, s_mm[0][w]), mx = max(mx, s_mm[1][w]);
    if (mx == 0) mn = 0;  // nothing visible
    kmin = (uint32_t)mn;
    const uint32_t span = (uint32_t)(mx - mn);
    sh = max(0, 32 - __clz((int)span) - DS_BITS);  // span < 2^(sh + 9)
    if (span == 0u) sh = 0;
    if (c == 0 && threadIdx.x == 0) range[2 * v] = kmin, range[2 * v + 1] = (uint32_t)sh;
  }
  __syncthreads();
  const int64_t vbase = sg.seg_off ? (int64_t)sg.seg_off[v] : (int64_t)v * P;
  const int n = !FIRST ? nvalid[v] : sg.seg_off ? sg.seg_off[v + 1] - sg.seg_off[v] : P;
  if (c * DS_CHUNK >= n) {  // (a short or empty segment: nothing of it in this chunk -- and nothing to read)
    uint16_t* dst0 = hist + ((int64_t)v * nchunk + c) * DS_BINS;
    for (int d = threadIdx.x; d < DS_BINS; d += DS_T) dst0[d] = 0;
    return;
  }
  // all loads first, on clamped indices and without branches (the compiler keeps a conditional load behind the LDS atomic
  // of the previous step: eight memory round trips per workgroup instead of one), then the histogram
  uint32_t dg[DS_STEPS];
  bool have[DS_STEPS];
#pragma unroll
  for (int it = 0; it < DS_STEPS; ++it) {
    const int t = c * DS_CHUNK + it * DS_T + threadIdx.x;
    const int tc = max(min(t, n - 1), 0);
    if (FIRST) {
      const uint32_t f = field[vbase + tc];
      have[it] = t < n && f != 0u;
      dg[it] = MSD ? ((f - kmin) >> sh) & dmask : f & dmask;  // (the mask only matters for culled entries: never counted)
    } else {
      have[it] = t < n;
      dg[it] = (uint32_t)(keys[vbase + tc] >> word_shift) & dmask;
    }
  }
#pragma unroll
  for (int it = 0; it < DS_STEPS; ++it)
    if (have[it]) atomicAdd(&s_h[dg[it]], 1u);
  __syncthreads();
  uint16_t* dst = hist + ((int64_t)v * nchunk + c) * DS_BINS;
  for (int d = threadIdx.x; d < DS_BINS; d += DS_T) dst[d] = (uint16_t)s_h[d];  // <= DS_CHUNK
}

// per (view, digit): exclusive prefix over the chunks.  A workgroup owns 64 digits of one view; its sixteen waves split the
// chunk range (two sweeps over the u16 table: sums, then prefixes), lanes = adjacent digits (128-byte rows).
constexpr int DS_SCAN_NW = 16;  // waves per scan workgroup
__global__ __launch_bounds__(DS_SCAN_NW* WAVE) void ds_scan_kernel(int nchunk, const uint16_t* __restrict__ hist,
                                                                   uint32_t* __restrict__ offs,
                                                                   int32_t* __restrict__ digit_total,
                                                                   int32_t* __restrict__ clear, int n_clear, int bins) {
  __shared__ unsigned int s_part[DS_SCAN_NW][WAVE];
  // (bucket path: the chunk totals the bucket launch adds into)
  if (clear != nullptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_clear; i += DS_SCAN_NW * WAVE) clear[i] = 0;
  const int v = blockIdx.x / (DS_BINS / WAVE), dg = blockIdx.x % (DS_BINS / WAVE);
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int d = dg * WAVE + lane;
  if (dg * WAVE >= bins) {  // (segments sorted on fewer bits: no key carries these digits; only their zero totals are read)
    if (wv == 0) digit_total[v * DS_BINS + d] = 0;
    return;
  }
  const int per = (nchunk + DS_SCAN_NW - 1) / DS_SCAN_NW;
  const int c0 = wv * per, c1 = min(nchunk, c0 + per);
  const uint16_t* col = hist + (int64_t)v * nchunk * DS_BINS + d;
  uint32_t* ocol = offs + (int64_t)v * nchunk * DS_BINS + d;
  constexpr int KEEP = 32;  // a wave's share of the chunks, kept in registers between the two sweeps when it fits
  unsigned int sum = 0;
  unsigned int run = 0;
  if (per <= KEEP) {
    // one trip to memory: all loads up front (clamped, unconditional), the prefix comes out of registers
    unsigned int val[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int c = min(c0 + i, nchunk - 1);
      val[i] = col[(int64_t)c * DS_BINS];
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i) sum += c0 + i < c1 ? val[i] : 0u;
    s_part[wv][lane] = sum;
    __syncthreads();
    for (int w = 0; w < wv; ++w) run += s_part[w][lane];
#pragma unroll
    for (int i = 0; i < KEEP; ++i)
      if (c0 + i < c1) {
        ocol[(int64_t)(c0 + i) * DS_BINS] = run;
        run += val[i];
      }
  } else {
#pragma unroll 8
    for (int c = c0; c < c1; ++c) sum += col[(int64_t)c * DS_BINS];  // (unrolled: eight independent loads in flight)
    s_part[wv][lane] = sum;
    __syncthreads();
    for (int w = 0; w < wv; ++w) run += s_part[w][lane];
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
      const unsigned int n = col[(int64_t)c * DS_BINS];
      ocol[(int64_t)c * DS_BINS] = run;
      run += n;
    }
  }
  if (wv == DS_SCAN_NW - 1) digit_total[v * DS_BINS + d] = (int32_t)run;
}

// FIRST: raw 4-byte fields (compaction, word assembly).  LAST: writes the id and the rectangle.
// id_bits > 0: packed words [field >> 9 | id (id_bits) | rectangle (26)]; id_bits == 0: (field << 32 | id) words.
// word_shift: where this pass's digit sits in the word (passes after the first).
// MSD (FIRST, not LAST): digit = (field - kmin) >> s, the word keeps the s bits below it (range[v] = {kmin, s}).
template <bool FIRST, bool LAST, bool LANE_ORDERED, bool MSD = false>
__global__ __launch_bounds__(DS_T) void ds_scatter_kernel(int P, int V, int nchunk, const int32_t* __restrict__ nvalid,
                                                          const uint32_t* __restrict__ field,
                                                          const uint64_t* __restrict__ keys_in, int word_shift, int id_bits,
                                                          uint32_t dmask,
                                                          const uint32_t* __restrict__ offs,
                                                          const int32_t* __restrict__ digit_total,
                                                          uint64_t* __restrict__ keys_out, const uint32_t* __restrict__ rect_raw,
                                                          uint32_t* __restrict__ rect_out, int32_t* __restrict__ ids_out,
                                                          int32_t* __restrict__ nvalid_out,
                                                          const uint32_t* __restrict__ range, DepthSortSegments sg) {
  static_assert(!MSD || (FIRST && !LAST), "the top-digit pass is a first pass with another launch behind it");
  __shared__ unsigned int s_cnt[DS_NW][DS_BINS];  // per-wave digit counters -> per-wave bases
  __shared__ int s_delta[DS_BINS];                // global position of a digit's run minus its chunk-local start
  __shared__ int s_w[DS_T / WAVE];
  __shared__ int s_wt[DS_T / WAVE];               // wave sums of the view's digit totals
  __shared__ uint64_t s_key[DS_CHUNK];
  __shared__ unsigned short s_dig[FIRST ? DS_CHUNK : 1];
  int v, c;
  if (!ds_block(V, nchunk, v, c)) return;
  const int n = !FIRST ? nvalid[v] : sg.seg_off ? sg.seg_off[v + 1] - sg.seg_off[v] : P;
  if (c * DS_CHUNK >= n) return;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const uint32_t kmin = MSD ? range[2 * v] : 0u;
  const int sh = MSD ? (int)range[2 * v + 1] : 0;
  for (int i = threadIdx.x; i < DS_NW * DS_BINS; i += DS_T) (&s_cnt[0][0])[i] = 0u;
  __syncthreads();
  // ---- phase 1: wave w takes positions [w * 512, (w + 1) * 512) of the chunk, 64 at a time in order
  const int64_t vbase = sg.seg_off ? (int64_t)sg.seg_off[v] : (int64_t)v * P;
  const int wbase = c * DS_CHUNK + wv * (DS_CHUNK / DS_NW);
  uint64_t key[DS_STEPS];
  unsigned short digit[DS_STEPS];
  int rank[DS_STEPS];  // -1: no entry
  const unsigned long long lt = (1ull << lane) - 1ull;
  // all loads first, on clamped indices and without branches (see ds_count_kernel), then the ranking
  bool present[DS_STEPS];
  uint32_t fv[FIRST ? DS_STEPS : 1], rv[FIRST ? DS_STEPS : 1];
  if (FIRST) {
#pragma unroll
    for (int j = 0; j < DS_STEPS; ++j) {
      const int64_t o = vbase + min(wbase + j * WAVE + lane, n - 1);
      fv[j] = field[o];
      rv[j] = rect_raw[o];  // unconditional (a branch here would serialise the loads again)
    }
  }
#pragma unroll
  for (int j = 0; j < DS_STEPS; ++j) {
    const int t = wbase + j * WAVE + lane;
    const int tc = min(t, n - 1);  // n > c * DS_CHUNK >= 0 here
    if (FIRST) {
      const uint32_t f = fv[j];
      const uint32_t rct = rv[j];
      present[j] = t < n && f != 0u;
      if (MSD) {
        const uint32_t rel = f - kmin;
        digit[j] = (unsigned short)((rel >> sh) & dmask);
        key[j] = ((uint64_t)(rel & ((1u << sh) - 1u)) << (id_bits + 26)) | ((uint64_t)(uint32_t)t << 26) | rct;
      } else {
        digit[j] = (unsigned short)(f & dmask);
        key[j] = id_bits > 0 ? ((uint64_t)(f >> DS_BITS) << (id_bits + 26)) | ((uint64_t)(uint32_t)t << 26) | rct
                             : ((uint64_t)f << 32) | (uint32_t)t;
      }
    } else {
      key[j] = keys_in[vbase + tc];
      present[j] = t < n;
      digit[j] = (unsigned short)((uint32_t)(key[j] >> word_shift) & dmask);
    }
  }
#pragma unroll
  for (int j = 0; j < DS_STEPS; ++j) {
    rank[j] = -1;
    const bool have = present[j];
    const uint32_t d = digit[j];
    if (LANE_ORDERED) {
      if (have) rank[j] = (int)atomicAdd(&s_cnt[wv][d], 1u);
    } else {
      unsigned long long peers = __ballot(have);
      if (peers != 0ull) {
#pragma unroll
        for (int bit = 0; bit < DS_BITS; ++bit) {
          const bool one = (d >> bit) & 1u;
          const unsigned long long bal = __ballot(one);
          peers &= one ? bal : ~bal;
        }
        if (have) {
          const unsigned int base = s_cnt[wv][d];
          rank[j] = (int)(base + (unsigned int)__popcll(peers & lt));
          if ((peers >> lane) == 1ull) s_cnt[wv][d] = base + (unsigned int)__popcll(peers);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  __syncthreads();
  // ---- phase 2: chunk-local start of every digit (exclusive scan over 512 digit totals, 2 per thread), the waves'
  //      bases inside a digit's run, and the run's global position
  int total = 0;
  {
    const int d0 = 2 * threadIdx.x;
    unsigned int cw[2][DS_NW];
    int tot[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      tot[u] = 0;
#pragma unroll
      for (int w = 0; w < DS_NW; ++w) {
        cw[u][w] = s_cnt[w][d0 + u];
        tot[u] += (int)cw[u][w];
      }
    }
    const int mine = tot[0] + tot[1];
    const int incl = wave_incl_scan_add_dpp(mine);
    if (lane == WAVE - 1) s_w[wv] = incl;
    // the view's digit bases = exclusive scan of its 512 digit totals, done here by every workgroup (it used to be a
    // launch of its own: three launches per frame whose only work was 512 additions)
    const int32_t* trow = digit_total + v * DS_BINS;
    const int t0 = trow[d0], t1 = trow[d0 + 1];
    const int tincl = wave_incl_scan_add_dpp(t0 + t1);
    if (lane == WAVE - 1) s_wt[wv] = tincl;
    __syncthreads();
    int base = 0, tbase = 0, view_total = 0;
    for (int i = 0; i < wv; ++i) base += s_w[i], tbase += s_wt[i];
    for (int i = 0; i < DS_T / WAVE; ++i) total += s_w[i], view_total += s_wt[i];
    if (FIRST && c == 0 && threadIdx.x == 0 && nvalid_out) nvalid_out[v] = view_total;  // entries that survived the culling
    const int dbase[2] = {tbase + tincl - (t0 + t1), tbase + tincl - t1};
    int start = base + incl - mine;
    const uint32_t* orow = offs + ((int64_t)v * nchunk + c) * DS_BINS;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      s_delta[d0 + u] = (int)orow[d0 + u] + dbase[u] - start;
      unsigned int run = (unsigned int)start;
#pragma unroll
      for (int w = 0; w < DS_NW; ++w) {
        s_cnt[w][d0 + u] = run;
        run += cw[u][w];
      }
      start += tot[u];
    }
  }
  __syncthreads();
  // ---- phase 3: every key to its chunk-local sorted position
#pragma unroll
  for (int j = 0; j < DS_STEPS; ++j)
    if (rank[j] >= 0) {
      const unsigned int lp = s_cnt[wv][digit[j]] + (unsigned int)rank[j];
      s_key[lp] = key[j];
      if (FIRST) s_dig[lp] = digit[j];  // a packed word no longer holds the bits this pass sorts on
    }
  __syncthreads();
  // ---- phase 4: contiguous runs per digit leave as full lines
  for (int lp = threadIdx.x; lp < total; lp += DS_T) {
    const uint64_t k = s_key[lp];
    const uint32_t d = FIRST ? (uint32_t)s_dig[lp] : (uint32_t)(k >> word_shift) & dmask;
    const int64_t gp = vbase + s_delta[d] + lp;
    if (LAST) {
      const uint32_t id = id_bits > 0 ? (uint32_t)(k >> 26) & ((1u << id_bits) - 1u) : (uint32_t)k;
      ids_out[gp] = (int32_t)id;
      rect_out[gp] = id_bits > 0 ? (uint32_t)k & 0x3ffffffu : rect_raw[vbase + id];
    } else {
      keys_out[gp] = k;
    }
  }
}

// One workgroup per (view, bucket of the top-digit pass): the bucket's words -- a contiguous run of keys[], in id order --
// leave in (rest of the key, id) order as ids + rectangles.
// BT threads and room for BCAP items: the launch for the usual buckets, and one with small workgroups for callers with
// tens of thousands of small buckets (grid_subsample: ~400 points each -- 512 threads and 80 KB of LDS per bucket made the
// launch 190 us, all of it per-workgroup overhead); a workgroup only takes buckets of n_lo < n <= n_hi entries.
template <bool LANE_ORDERED, int BT, int BCAP>
__global__ __launch_bounds__(BT) void ds_bucket_sort_kernel(int P, int n_lo, int n_hi, const int32_t* __restrict__ digit_total,
                                                               const uint32_t* __restrict__ range,
                                                               const uint64_t* __restrict__ keys, int id_bits,
                                                               int32_t* __restrict__ ids_out, uint32_t* __restrict__ rect_out,
                                                               int32_t* __restrict__ flag, int flag_value, DepthSortTotals ct,
                                                               DepthSortSegments sg) {
  constexpr int BNW = BT / WAVE, BSTEPS = (BCAP + BT - 1) / BT, DPT = DS_BINS / BT;  // digits per thread in the scan
  static_assert(DS_BINS % BT == 0 && BCAP <= (1 << MSD_IDX_BITS), "bucket workgroup shape");
  __shared__ unsigned int s_cnt[BNW][DS_BINS];
  __shared__ uint32_t s_item[2][BCAP];
  __shared__ int s_w[BNW];
  __shared__ int s_ct[BCAP / 64 + 2];  // tile instances per chunk the bucket touches (chunk >= 64 entries)
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  auto one_bucket = [&](const int v, const int d) {
  const int32_t* trow = digit_total + v * DS_BINS;
  const int n = trow[d];
  if (n <= n_lo || n > n_hi) {  // (n_lo >= 0: empty buckets leave here)
    // two size classes: the launch of the small class lists the buckets of the large one (usually none)
    if (n > n_hi && sg.big_list != nullptr && threadIdx.x == 0) sg.big_list[atomicAdd(sg.big_cnt, 1)] = v * DS_BINS + d;
    return;
  }
  // where the bucket starts: the totals of the digits below
  {
    int mine = 0;
    for (int i = threadIdx.x; i < d; i += BT) mine += trow[i];
    const int ws = wave_sum_i32_dpp(mine);
    if (lane == 0) s_w[wv] = ws;
  }
  __syncthreads();
  int start = 0;
#pragma unroll
  for (int w = 0; w < BNW; ++w) start += s_w[w];
  const int sh = (int)range[2 * v + 1];
  const int64_t vbase = sg.seg_off ? (int64_t)sg.seg_off[v] : (int64_t)v * P;
  const int64_t base = vbase + start;
  const int32_t id_add = sg.seg_off ? sg.seg_off[v] : 0;  // segments with offsets: ids leave as positions in the whole array
  const uint32_t idmask = (1u << id_bits) - 1u;
  if (n > BCAP && sh != 0) {
    // the frame will be repeated; until then the launches behind this one must find nothing to bin here
    // (empty rectangles, no chunk totals) rather than whatever the buffers held
    if (threadIdx.x == 0) {
      if (flag_value > 0) atomicOr(flag, flag_value);  // a bit next to others
      else *flag = flag_value;                         // a per-call stamp
    }
    for (int i = threadIdx.x; i < n; i += BT) {
      ids_out[base + i] = id_add;
      if (sg.key64_out != nullptr) sg.key64_out[base + i] = (uint64_t)v << sg.key64_shift;
      else rect_out[base + i] = 0u;
    }
    return;
  }
  // one entry leaves: id, rectangle, and its tile instances into the total of the chunk its position falls into.  64
  // consecutive positions touch at most two chunks: two wave sums, then one LDS add each.
  const int chunk0 = ct.chunk_total != nullptr ? start / ct.chunk : 0;
  const bool few_chunks = ct.chunk_total != nullptr && (start + n - 1) / ct.chunk - chunk0 < (int)(sizeof(s_ct) / sizeof(int));
  if (ct.chunk_total != nullptr && few_chunks)
    for (int i = threadIdx.x; i < (int)(sizeof(s_ct) / sizeof(int)); i += BT) s_ct[i] = 0;
  auto emit = [&](int j, bool have, uint64_t k) {  // wave-uniform call; j = position in the bucket
    const uint32_t id = (uint32_t)(k >> 26) & idmask, r = (uint32_t)k & 0x3ffffffu;
    if (have) {
      ids_out[base + j] = (int32_t)id + id_add;
      if (sg.gather64 != nullptr) sg.key64_out[base + j] = sg.gather64[(int64_t)id + id_add];  // a 64-bit value by index
      else if (sg.key64_out != nullptr) sg.key64_out[base + j] = ((uint64_t)v << sg.key64_shift) | r;  // (segment, payload) words
      else rect_out[base + j] = r;
    }
    if (ct.chunk_total != nullptr) {
      int x0, y0, w = 0, h = 0;
      if (have && r != 0u) rect_decode(r, (int)id, vbase, ct.rec, ct.gx, ct.gy, x0, y0, w, h);
      const int nt = have ? w * h : 0;
      const int ck = (start + j) / ct.chunk;
      const int ck_first = __builtin_amdgcn_readfirstlane(ck);
      const int a = wave_sum_i32_dpp(ck == ck_first ? nt : 0), b = wave_sum_i32_dpp(ck == ck_first ? 0 : nt);
      if (lane == 0) {
        if (few_chunks) {
          if (a) atomicAdd(&s_ct[ck_first - chunk0], a);
          if (b) atomicAdd(&s_ct[ck_first + 1 - chunk0], b);
        } else {  // (a bucket of equal keys longer than the table covers: straight to memory)
          if (a) atomicAdd(&ct.chunk_total[v * ct.nchunk + ck_first], a);
          if (b) atomicAdd(&ct.chunk_total[v * ct.nchunk + ck_first + 1], b);
        }
      }
    }
  };
  auto flush_totals = [&]() {
    if (ct.chunk_total == nullptr || !few_chunks) return;
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(sizeof(s_ct) / sizeof(int)); i += BT)
      if (s_ct[i] != 0) atomicAdd(&ct.chunk_total[v * ct.nchunk + chunk0 + i], s_ct[i]);
  };
  if (sh == 0 || n == 1) {  // equal keys: id order is the order
    __syncthreads();  // (s_ct cleared)
    for (int i0 = 0; i0 < n; i0 += BT) {
      const int i = i0 + (int)threadIdx.x;
      if (i0 + wv * WAVE >= n) break;  // wave-uniform
      emit(i, i < n, i < n ? keys[base + i] : 0ull);
    }
    flush_totals();
    return;
  }
  for (int i = threadIdx.x; i < n; i += BT)
    s_item[0][i] = ((uint32_t)(keys[base + i] >> (id_bits + 26)) << MSD_IDX_BITS) | (uint32_t)i;
  // a wave's share of the bucket: whole 64-item steps
  const int seg = ((n + BNW - 1) / BNW + WAVE - 1) / WAVE * WAVE;
  const int w0 = wv * seg, w1 = min(n, w0 + seg);
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cur = 0;
  for (int shift = MSD_IDX_BITS; shift < MSD_IDX_BITS + sh; shift += DS_BITS) {
    const uint32_t dmask = (1u << min(DS_BITS, MSD_IDX_BITS + sh - shift)) - 1u;
    for (int i = threadIdx.x; i < BNW * DS_BINS; i += BT) (&s_cnt[0][0])[i] = 0u;
    __syncthreads();  // (also: the items of the previous round are in place)
    const uint32_t* src = s_item[cur];
    uint32_t* dst = s_item[cur ^ 1];
    int rank[BSTEPS];
#pragma unroll
    for (int j = 0; j < BSTEPS; ++j) {
      rank[j] = -1;
      if (w0 + j * WAVE >= w1) continue;  // wave-uniform
      const int i = w0 + j * WAVE + lane;
      const bool have = i < w1;
      const uint32_t dg = have ? (src[i] >> shift) & dmask : 0u;
      if (LANE_ORDERED) {
        if (have) rank[j] = (int)atomicAdd(&s_cnt[wv][dg], 1u);
      } else {
        unsigned long long peers = __ballot(have);
#pragma unroll
        for (int bit = 0; bit < DS_BITS; ++bit) {
          const bool one = (dg >> bit) & 1u;
          const unsigned long long bal = __ballot(one);
          peers &= one ? bal : ~bal;
        }
        if (have) {
          const unsigned int b0 = s_cnt[wv][dg];
          rank[j] = (int)(b0 + (unsigned int)__popcll(peers & lt));
          if ((peers >> lane) == 1ull) s_cnt[wv][dg] = b0 + (unsigned int)__popcll(peers);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    // digit starts (a thread takes DPT adjacent digits) and the waves' bases inside a digit's run
    {
      unsigned int cw[DPT][BNW];
      int tot = 0;
#pragma unroll
      for (int u = 0; u < DPT; ++u)
#pragma unroll
        for (int w = 0; w < BNW; ++w) {
          cw[u][w] = s_cnt[w][threadIdx.x * DPT + u];
          tot += (int)cw[u][w];
        }
      const int incl = wave_incl_scan_add_dpp(tot);
      if (lane == WAVE - 1) s_w[wv] = incl;
      __syncthreads();
      int run = incl - tot;
      for (int w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
      for (int u = 0; u < DPT; ++u)
#pragma unroll
        for (int w = 0; w < BNW; ++w) {
          s_cnt[w][threadIdx.x * DPT + u] = (unsigned int)run;
          run += (int)cw[u][w];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BSTEPS; ++j)
      if (rank[j] >= 0) {
        const uint32_t it = src[w0 + j * WAVE + lane];
        dst[s_cnt[wv][(it >> shift) & dmask] + (unsigned int)rank[j]] = it;
      }
    cur ^= 1;
    __syncthreads();  // the bases are read to the end before the next round clears them; the items are in place
  }
  for (int j0 = 0; j0 < n; j0 += BT) {
    const int j = j0 + (int)threadIdx.x;
    if (j0 + wv * WAVE >= n) break;  // wave-uniform
    emit(j, j < n, j < n ? keys[base + (s_item[cur][j] & ((1u << MSD_IDX_BITS) - 1u))] : 0ull);
  }
  flush_totals();
  };  // one_bucket
  if (n_lo > 0 && sg.big_list != nullptr) {
    // the large class of a two-class call: a fixed grid walks the list the small launch left
    const int listed = *sg.big_cnt;
    for (int i = blockIdx.y * gridDim.x + blockIdx.x; i < listed; i += gridDim.x * gridDim.y) {
      const int code = sg.big_list[i];
      one_bucket(code / DS_BINS, code % DS_BINS);
      __syncthreads();  // (the LDS arrays are the next bucket's)
    }
    return;
  }
  if (sg.xcd_segments > 0) {
    // many segments: all buckets of a segment on one XCD -- what a bucket's entries look up by index (gather64: a value per
    // entry at a random place of the segment's own array) is then fetched by one L2, not by eight
    const int id = (int)blockIdx.x, slot = id >> 3, bins = sg.bins_used, g = slot / bins;
    const int v = g * 8 + (id & 7);
    if (v < sg.xcd_segments) one_bucket(v, slot - g * bins);
    return;
  }
  one_bucket((int)blockIdx.y, (int)blockIdx.x);
}

}  // namespace

size_t depth_sort_table_bytes(int64_t P, int V) {
  const int64_t nchunk = (P + DS_CHUNK - 1) / DS_CHUNK;
  return align_up((size_t)V * nchunk * DS_BINS * sizeof(uint16_t), 256) +
         align_up((size_t)V * nchunk * DS_BINS * sizeof(uint32_t), 256) + align_up((size_t)V * DS_BINS * sizeof(int32_t), 256) +
         align_up((size_t)V * 2 * sizeof(uint32_t), 256) + align_up(((size_t)V * DS_BINS + 1) * sizeof(int32_t), 256) + 256;
}

bool depth_sort_msd_possible(int64_t P, int V, int key_bits) {
  // the word holds [<= 18 rest bits | id | 26 rectangle bits]; a few views per call only: with 32 the launches are not the
  // cost, and two passes + the bucket launch measured no faster than three passes (0.433 vs 0.440 ms) while the preprocess
  // paid 0.06 ms for the per-view key ranges
  return V >= 1 && V <= 4 && P <= (1ll << 20) && key_bits <= 2 * DS_BITS + DS_BITS;
}

// field, rect_raw: [V*P] per (view, Gaussian), left untouched.  keys_a, keys_b: [V*P] scratch.  Out: ids_out / rect_out
// [V*P] (the first nvalid_out[v] entries of every view's stride-P segment), nvalid_out [V] on the device.
int depth_sort_views(const uint32_t* field, const uint32_t* rect_raw, uint64_t* keys_a, uint64_t* keys_b, int32_t* ids_out,
                     uint32_t* rect_out, int32_t* nvalid_out, int64_t P, int V, int key_bits, void* table, size_t table_bytes,
                     hipStream_t stream, const int2* key_mm, int nb_mm, int32_t* overflow_flag, int overflow_value,
                     const DepthSortTotals* chunk_totals, const DepthSortSegments* segments) {
  if (P <= 0 || V <= 0) return GR_OK;
  GR_REQUIRE(key_bits >= 1 && key_bits <= 32, "depth_sort: key_bits %d out of range", key_bits);
  GR_REQUIRE(table && table_bytes >= depth_sort_table_bytes(P, V), "depth_sort: table too small");
  const int nchunk = (int)((P + DS_CHUNK - 1) / DS_CHUNK);
  Carver cv(table);
  uint16_t* hist = cv.take<uint16_t>((size_t)V * nchunk * DS_BINS);
  uint32_t* offs = cv.take<uint32_t>((size_t)V * nchunk * DS_BINS);
  int32_t* dbase = cv.take<int32_t>((size_t)V * DS_BINS);
  uint32_t* range = cv.take<uint32_t>((size_t)V * 2);
  int32_t* big = cv.take<int32_t>((size_t)V * DS_BINS + 1);  // [0] count, then the listed (view, bucket) codes
  bool ordered = false;
  int rc = lds_atomics_lane_ordered(stream, &ordered);
  if (rc != GR_OK) return rc;
  const int passes = (key_bits + DS_BITS - 1) / DS_BITS;
  const dim3 grid((unsigned)ds_grid(V, nchunk)), blk(DS_T);
  int id_bits = 1;
  while (id_bits < 32 && (1ll << id_bits) < P) ++id_bits;
  DepthSortSegments sg{nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
  if (segments != nullptr) sg = *segments;
  if (key_mm != nullptr || segments != nullptr) {  // top-digit pass + one launch over the buckets (see the head of this file)
    GR_REQUIRE(overflow_flag != nullptr && P <= (1ll << 20) && key_bits <= 3 * DS_BITS &&
                   (segments != nullptr ? (key_mm == nullptr && sg.range_in != nullptr)
                                        : (depth_sort_msd_possible(P, V, key_bits) && nb_mm > 0)),
               "depth_sort: bucket path misused");
    const uint32_t dmask = DS_BINS - 1;
    if (segments != nullptr) range = const_cast<uint32_t*>(sg.range_in);
    hipLaunchKernelGGL((ds_count_kernel<true, true>), grid, blk, 0, stream, (int)P, V, nchunk, nvalid_out, field,
                       (const uint64_t*)nullptr, 0, dmask, hist, key_mm, nb_mm, range, sg);
    DepthSortTotals ct{nullptr, 1, 0, nullptr, 0, 0};
    if (chunk_totals != nullptr) ct = *chunk_totals;
    // many small buckets (segments: tens of thousands of a few hundred entries): those first, on small workgroups, which
    // also list the few large ones for a launch of a fixed grid (a workgroup per bucket that only finds out that the
    // bucket is not its size cost 22 us at 32 768 buckets)
    constexpr int SMALL_T = 128, SMALL_CAP = 1024;
    const bool two_sizes = segments != nullptr;
    const int bins_used = (two_sizes && sg.bins_used > 0 && sg.bins_used < DS_BINS) ? sg.bins_used : DS_BINS;
    GR_REQUIRE(!(two_sizes && chunk_totals != nullptr), "depth_sort: chunk totals and segments together");
    if (two_sizes) sg.big_cnt = big, sg.big_list = big + 1;
    hipLaunchKernelGGL(ds_scan_kernel, dim3((unsigned)(V * (DS_BINS / WAVE))), dim3(DS_SCAN_NW * WAVE), 0, stream, nchunk, hist,
                       offs, dbase, two_sizes ? big : ct.chunk_total, two_sizes ? 1 : V * ct.nchunk, bins_used);
    if (ordered)
      hipLaunchKernelGGL((ds_scatter_kernel<true, false, true, true>), grid, blk, 0, stream, (int)P, V, nchunk, nvalid_out, field,
                         (const uint64_t*)nullptr, 0, id_bits, dmask, offs, dbase, keys_a, rect_raw, rect_out, ids_out, nvalid_out,
                         range, sg);
    else
      hipLaunchKernelGGL((ds_scatter_kernel<true, false, false, true>), grid, blk, 0, stream, (int)P, V, nchunk, nvalid_out, field,
                         (const uint64_t*)nullptr, 0, id_bits, dmask, offs, dbase, keys_a, rect_raw, rect_out, ids_out, nvalid_out,
                         range, sg);
    dim3 bgrid((unsigned)bins_used, (unsigned)V);
    if (two_sizes && V >= 32) {
      sg.bins_used = bins_used;
      sg.xcd_segments = V;
      bgrid = dim3((unsigned)(bins_used * ((V + 7) / 8 * 8)), 1);
    }
    const int split = two_sizes ? SMALL_CAP : 0;
    const dim3 biggrid = two_sizes ? dim3(std::min<unsigned>(1024u, (unsigned)V * DS_BINS), 1) : bgrid;
    if (two_sizes) {
      if (ordered)
        hipLaunchKernelGGL((ds_bucket_sort_kernel<true, SMALL_T, SMALL_CAP>), bgrid, dim3(SMALL_T), 0, stream, (int)P, 0, split, dbase,
                           range, keys_a, id_bits, ids_out, rect_out, overflow_flag, overflow_value, ct, sg);
      else
        hipLaunchKernelGGL((ds_bucket_sort_kernel<false, SMALL_T, SMALL_CAP>), bgrid, dim3(SMALL_T), 0, stream, (int)P, 0, split, dbase,
                           range, keys_a, id_bits, ids_out, rect_out, overflow_flag, overflow_value, ct, sg);
    }
    if (ordered)
      hipLaunchKernelGGL((ds_bucket_sort_kernel<true, MSD_T, MSD_CAP>), biggrid, dim3(MSD_T), 0, stream, (int)P, split, 0x7fffffff, dbase,
                         range, keys_a, id_bits, ids_out, rect_out, overflow_flag, overflow_value, ct, sg);
    else
      hipLaunchKernelGGL((ds_bucket_sort_kernel<false, MSD_T, MSD_CAP>), biggrid, dim3(MSD_T), 0, stream, (int)P, split, 0x7fffffff, dbase,
                         range, keys_a, id_bits, ids_out, rect_out, overflow_flag, overflow_value, ct, sg);
    GR_LAUNCH_CHECK();
    return GR_OK;
  }
  const int rest_bits = std::max(0, key_bits - DS_BITS);  // field bits still to be sorted after the first pass
  const bool packed = rest_bits + id_bits + 26 <= 64;
  if (!packed) id_bits = 0;
  uint64_t* kbuf[2] = {keys_a, keys_b};
  for (int p = 0; p < passes; ++p) {
    const int shift = p * DS_BITS;
    const int word_shift = packed ? id_bits + 26 + (p - 1) * DS_BITS : 32 + shift;  // passes after the first
    const int nb = std::min(DS_BITS, key_bits - shift);
    const uint32_t dmask = (1u << nb) - 1u;
    const bool first = p == 0, last = p == passes - 1;
    const uint64_t* kin = first ? nullptr : kbuf[(p + 1) % 2];
    uint64_t* kout = kbuf[p % 2];
    if (first)
      hipLaunchKernelGGL((ds_count_kernel<true, false>), grid, blk, 0, stream, (int)P, V, nchunk, nvalid_out, field, kin,
                         word_shift, dmask, hist, (const int2*)nullptr, 0, (uint32_t*)nullptr, sg);
    else
      hipLaunchKernelGGL((ds_count_kernel<false, false>), grid, blk, 0, stream, (int)P, V, nchunk, nvalid_out, field, kin,
                         word_shift, dmask, hist, (const int2*)nullptr, 0, (uint32_t*)nullptr, sg);
    hipLaunchKernelGGL(ds_scan_kernel, dim3((unsigned)(V * (DS_BINS / WAVE))), dim3(DS_SCAN_NW * WAVE), 0, stream, nchunk, hist,
                       offs, dbase, (int32_t*)nullptr, 0, DS_BINS);
#define GR_DS_SCATTER(F, L, O)                                                                                            \
  hipLaunchKernelGGL((ds_scatter_kernel<F, L, O>), grid, blk, 0, stream, (int)P, V, nchunk, nvalid_out, field, kin, \
                     word_shift, id_bits, dmask, offs, dbase, kout, rect_raw, rect_out, ids_out, first ? nvalid_out : nullptr, \
                     (const uint32_t*)nullptr, sg)
    if (ordered) {
      if (first && last) GR_DS_SCATTER(true, true, true);
      else if (first) GR_DS_SCATTER(true, false, true);
      else if (last) GR_DS_SCATTER(false, true, true);
      else GR_DS_SCATTER(false, false, true);
    } else {
      if (first && last) GR_DS_SCATTER(true, true, false);
      else if (first) GR_DS_SCATTER(true, false, false);
      else if (last) GR_DS_SCATTER(false, true, false);
      else GR_DS_SCATTER(false, false, false);
    }
#undef GR_DS_SCATTER
    GR_LAUNCH_CHECK();
  }
  return GR_OK;
}

}  // namespace gr
