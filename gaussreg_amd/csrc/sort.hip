// Device-wide STABLE LSD radix sort of (u64 key, i32 value) pairs on bits [begin_bit, end_bit) -- this library's own
// kernels (rounds 1-2 passed through to rocPRIM here).  Users: grid_subsample's (cloud, voxel key) sort and the Morton
// pre-sort of the farthest point sampling.
//
// One pass per digit of up to 11 bits (the digit width is chosen so that all passes are equally wide):
//   rs_hist_kernel     per-chunk digit histogram (2048 keys per workgroup, LDS atomics) -> hist[digit][chunk]
//   rs_scan_kernel     one wave per digit: exclusive prefix over the chunks of the digit (contiguous; in place) and the
//                      digit totals
//   rs_scatter_kernel  every workgroup scans the digit totals itself (<= 2048 values), then ranks its chunk: wave w owns a
//                      contiguous quarter of the chunk and a private row of running digit counters in LDS; keys are
//                      taken 64 at a time in position order and ranked among the equal-digit lanes of the step by
//                      explicit ballots (one per digit bit) -- stable by construction, no reliance on the lane order of
//                      LDS atomics.  Three launches per pass.
#include <algorithm>

#include "common.hpp"

namespace gr {
namespace {

constexpr int RS_T = 256;                 // threads per workgroup
constexpr int RS_PER = 8;                 // keys per thread
constexpr int RS_CHUNK = RS_T * RS_PER;   // 2048 keys per workgroup
constexpr int RS_NW = RS_T / WAVE;        // 4 waves
constexpr int RS_WKEYS = RS_CHUNK / RS_NW;  // 512 keys per wave, in position order
constexpr int RS_MAX_BITS = 11;
constexpr int RS_MAX_BINS = 1 << RS_MAX_BITS;

__device__ __forceinline__ unsigned digit_of(uint64_t k, int shift, unsigned mask) { return (unsigned)(k >> shift) & mask; }

__global__ __launch_bounds__(RS_T) void rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, unsigned mask,
                                                       int bins, uint32_t* __restrict__ hist) {
  extern __shared__ unsigned int s_h[];
  for (int i = threadIdx.x; i < bins; i += RS_T) s_h[i] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
  uint64_t k[RS_PER];
#pragma unroll
  for (int u = 0; u < RS_PER; ++u) k[u] = keys[min(base + u * RS_T + threadIdx.x, n - 1)];  // all loads first
#pragma unroll
  for (int u = 0; u < RS_PER; ++u)
    if (base + u * RS_T + threadIdx.x < n) atomicAdd(&s_h[digit_of(k[u], shift, mask)], 1u);
  __syncthreads();
  // digit-major: hist[d][chunk] -- the scan over the chunks of a digit reads contiguous words
  for (int i = threadIdx.x; i < bins; i += RS_T) hist[(int64_t)i * gridDim.x + blockIdx.x] = s_h[i];
}

// one wave per digit: exclusive prefix over the chunks of that digit (contiguous in the digit-major table), in place
__global__ __launch_bounds__(RS_T) void rs_scan_kernel(uint32_t* __restrict__ hist, int nchunk, int bins,
                                                       uint32_t* __restrict__ totals) {
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const int d = blockIdx.x * RS_NW + w;
  if (d >= bins) return;
  uint32_t* row = hist + (int64_t)d * nchunk;
  unsigned int carry = 0u;
  for (int c0 = 0; c0 < nchunk; c0 += WAVE) {
    const int c = c0 + lane;
    const unsigned int v = c < nchunk ? row[c] : 0u;
    const unsigned int inc = (unsigned int)wave_incl_scan_add_dpp((int)v);
    if (c < nchunk) row[c] = carry + inc - v;
    carry += (unsigned int)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
  }
  if (lane == 0) totals[d] = carry;
}

// the same for long tables (millions of keys: thousands of chunks per digit): one workgroup per digit, 2048 chunks per step
// as two int4 per thread, block scan, running carry -- a single wave walking 6 250 chunks 64 at a time took 37 us per pass
__global__ __launch_bounds__(RS_T) void rs_scan_long_kernel(uint32_t* __restrict__ hist, int nchunk, uint32_t* __restrict__ totals) {
  __shared__ unsigned int s_w[RS_NW];
  const int d = blockIdx.x, lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  uint32_t* row = hist + (int64_t)d * nchunk;
  unsigned int carry = 0u;
  for (int c0 = 0; c0 < nchunk; c0 += RS_T * 8) {
    const int base = c0 + (int)threadIdx.x * 8;
    unsigned int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base + k < nchunk ? row[base + k] : 0u;
    unsigned int sum = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += v[k];
    const unsigned int inc = (unsigned int)wave_incl_scan_add_dpp((int)sum);
    if (lane == WAVE - 1) s_w[w] = inc;
    __syncthreads();
    unsigned int ex = carry + inc - sum, tot = 0u;
#pragma unroll
    for (int i = 0; i < RS_NW; ++i) {
      ex += i < w ? s_w[i] : 0u;
      tot += s_w[i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (base + k < nchunk) row[base + k] = ex;
      ex += v[k];
    }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[d] = carry;
}

// STAGE: the chunk is first sorted inside LDS and leaves as per-digit runs (consecutive keys of a digit go to consecutive
// addresses: with 8-bit digits a run is 8 keys = one 64-byte sector).  Without it every key is one scattered 8 + 4 byte
// write -- fine for the launch-bound sizes (<= 500 k keys, 11-bit digits: a run would be one key anyway), 2-3x slower at
// millions of keys.
template <bool IOTA, bool STAGE>
__global__ __launch_bounds__(RS_T) void rs_scatter_kernel(const uint64_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                                                          int64_t iota_period, int64_t n, int shift, int nbits, int bins,
                                                          const uint32_t* __restrict__ offs, const uint32_t* __restrict__ totals,
                                                          uint64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned int s_mem[];
  unsigned int* s_dbase = s_mem;               // [bins] exclusive prefix of the digit totals
  unsigned int* s_cnt = s_mem + bins;          // [RS_NW][bins] per-wave counts, then running destinations
  // STAGE: [bins] global destination minus local position per digit, then the staged chunk
  unsigned int* s_delta = s_cnt + RS_NW * bins;
  uint64_t* s_keys = reinterpret_cast<uint64_t*>(s_delta + bins + (bins & 1));
  int32_t* s_vals = reinterpret_cast<int32_t*>(s_keys + RS_CHUNK);
  __shared__ unsigned int s_wsum[RS_NW];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
  const unsigned mask = (1u << nbits) - 1u;
  const int per = (bins + RS_T - 1) / RS_T;
  // ---- digit bases: scan of <= 2048 totals (8 per thread)
  {
    unsigned int loc[RS_MAX_BINS / RS_T];
    unsigned int sum = 0u;
#pragma unroll
    for (int u = 0; u < RS_MAX_BINS / RS_T; ++u) {
      const int d = tid * per + u;
      loc[u] = (u < per && d < bins) ? totals[d] : 0u;
      sum += loc[u];
    }
    const unsigned int inc = (unsigned int)wave_incl_scan_add_dpp((int)sum);
    if (lane == WAVE - 1) s_wsum[w] = inc;
    for (int i = tid; i < RS_NW * bins; i += RS_T) s_cnt[i] = 0u;
    __syncthreads();
    unsigned int ex = inc - sum;
    for (int i = 0; i < w; ++i) ex += s_wsum[i];
#pragma unroll
    for (int u = 0; u < RS_MAX_BINS / RS_T; ++u) {
      const int d = tid * per + u;
      if (u < per && d < bins) s_dbase[d] = ex;
      ex += loc[u];
    }
  }
  // ---- the wave's 512 keys in position order: step r, lane l <-> position chunk + 512 w + 64 r + l
  const int64_t cbase = (int64_t)blockIdx.x * RS_CHUNK;
  const int64_t wbase = cbase + (int64_t)w * RS_WKEYS;
  uint64_t k[RS_PER];
  int32_t v[RS_PER];
#pragma unroll
  for (int r = 0; r < RS_PER; ++r) {
    const int64_t p = min(wbase + r * WAVE + lane, n - 1);
    k[r] = keys_in[p];
    v[r] = IOTA ? (int32_t)(p % iota_period) : vals_in[p];
  }
  unsigned int* row = s_cnt + w * bins;
#pragma unroll
  for (int r = 0; r < RS_PER; ++r)
    if (wbase + r * WAVE + lane < n) atomicAdd(&row[digit_of(k[r], shift, mask)], 1u);
  __syncthreads();
  const uint32_t* off_c = offs + blockIdx.x;  // digit-major table: entry of digit d at off_c[d * gridDim.x]
  const int64_t ostr = gridDim.x;
  if (STAGE) {
    // local exclusive prefix over the digits of the chunk (thread t owns digits t*per .. t*per+per-1, like the scan above)
    unsigned int loc[RS_MAX_BINS / RS_T];
    unsigned int sum = 0u;
#pragma unroll
    for (int u = 0; u < RS_MAX_BINS / RS_T; ++u) {
      const int d = tid * per + u;
      unsigned int c = 0u;
      if (u < per && d < bins)
        for (int i = 0; i < RS_NW; ++i) c += s_cnt[i * bins + d];
      loc[u] = c;
      sum += c;
    }
    const unsigned int inc = (unsigned int)wave_incl_scan_add_dpp((int)sum);
    __syncthreads();  // s_wsum is reused
    if (lane == WAVE - 1) s_wsum[w] = inc;
    __syncthreads();
    unsigned int ex = inc - sum;
    for (int i = 0; i < w; ++i) ex += s_wsum[i];
#pragma unroll
    for (int u = 0; u < RS_MAX_BINS / RS_T; ++u) {
      const int d = tid * per + u;
      if (u < per && d < bins) {
        s_delta[d] = s_dbase[d] + off_c[d * ostr] - ex;  // global destination of the digit's first key minus its local position
        unsigned int acc = ex;                      // running LOCAL destinations per wave
        for (int i = 0; i < RS_NW; ++i) {
          const unsigned int c = s_cnt[i * bins + d];
          s_cnt[i * bins + d] = acc;
          acc += c;
        }
      }
      ex += loc[u];
    }
  } else {
    // ---- running destinations: row[w][d] = global offset of (chunk, d) + digit base + keys of earlier waves with digit d
    for (int d = tid; d < bins; d += RS_T) {
      unsigned int acc = s_dbase[d] + off_c[d * ostr];
#pragma unroll
      for (int i = 0; i < RS_NW; ++i) {
        const unsigned int c = s_cnt[i * bins + d];
        s_cnt[i * bins + d] = acc;
        acc += c;
      }
    }
  }
  __syncthreads();
  // ---- stable ranking, 64 keys per step: lanes with my digit (one ballot per digit bit), my rank among them
#pragma unroll
  for (int r = 0; r < RS_PER; ++r) {
    const bool live = wbase + r * WAVE + lane < n;
    const unsigned d = digit_of(k[r], shift, mask);
    unsigned long long same = __ballot(live);
    for (int b = 0; b < nbits; ++b) {
      const unsigned long long mb = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? mb : ~mb;
    }
    if (live) {
      const unsigned long long below = same & ((1ull << lane) - 1ull);
      const unsigned int dst = row[d] + (unsigned int)__popcll(below);
      if (STAGE) {
        s_keys[dst] = k[r];
        s_vals[dst] = v[r];
      } else {
        keys_out[dst] = k[r];
        vals_out[dst] = v[r];
      }
      if ((same >> lane) >> 1 == 0ull) row[d] += (unsigned int)__popcll(same);  // the group's last lane moves the counter on
    }
    // (a wave executes in lockstep and the counter of a digit is touched by one lane per step: no barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (STAGE) {
    __syncthreads();
    const int here = (int)min((int64_t)RS_CHUNK, n - cbase);
    for (int i = tid; i < here; i += RS_T) {
      const uint64_t key = s_keys[i];
      const unsigned int dst = (unsigned int)i + s_delta[digit_of(key, shift, mask)];
      keys_out[dst] = key;
      vals_out[dst] = s_vals[i];
    }
  }
}

struct Plan {
  int passes, digit_bits, nchunk;
  bool stage;
};

Plan plan_of(int64_t n, int begin_bit, int end_bit) {
  Plan p;
  const int bits = end_bit - begin_bit;
  // launch-bound sizes: as few passes as possible (11-bit digits); millions of keys: 8-bit digits whose runs are whole
  // sectors, written out of an LDS-sorted chunk
  p.stage = n >= (1 << 19);
  const int max_bits = p.stage ? 8 : RS_MAX_BITS;
  p.passes = (bits + max_bits - 1) / max_bits;
  p.digit_bits = (bits + p.passes - 1) / p.passes;
  p.nchunk = (int)((n + RS_CHUNK - 1) / RS_CHUNK);
  return p;
}

int radix_sort(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const int32_t* vals_in,
               int64_t iota_period, int32_t* vals_out, int64_t n, int begin_bit, int end_bit, hipStream_t stream) {
  GR_REQUIRE(begin_bit >= 0 && end_bit <= 64 && end_bit > begin_bit, "radix sort: bad bit range [%d, %d)", begin_bit, end_bit);
  GR_REQUIRE(n < (1ll << 31), "radix sort: too many keys");
  GR_REQUIRE(temp != nullptr && temp_bytes >= sort_pairs_temp_bytes(n), "sort temp storage too small");
  const Plan pl = plan_of(n, begin_bit, end_bit);
  // temp: [keys ping | vals ping | hist (nchunk x bins) | totals (bins)]  (an odd number of passes ends in keys_out either way)
  Carver cv(temp);
  uint64_t* kt = cv.take<uint64_t>(n);
  int32_t* vt = cv.take<int32_t>(n);
  uint32_t* hist = cv.take<uint32_t>((size_t)pl.nchunk * RS_MAX_BINS);
  uint32_t* totals = cv.take<uint32_t>(RS_MAX_BINS);
  // ping-pong so that the LAST pass writes keys_out / vals_out
  const uint64_t* kin = keys_in;
  const int32_t* vin = vals_in;
  for (int p = 0; p < pl.passes; ++p) {
    const int shift = begin_bit + p * pl.digit_bits;
    const int nb = std::min(pl.digit_bits, end_bit - shift);
    const int bins = 1 << nb;
    const bool to_out = ((pl.passes - 1 - p) % 2) == 0;
    uint64_t* ko = to_out ? keys_out : kt;
    int32_t* vo = to_out ? vals_out : vt;
    const bool iota = p == 0 && iota_period > 0;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(pl.nchunk), dim3(RS_T), bins * sizeof(unsigned int), stream, kin, n, shift,
                       (unsigned)(bins - 1), bins, hist);
    if (pl.nchunk > 1024)
      hipLaunchKernelGGL(rs_scan_long_kernel, dim3(bins), dim3(RS_T), 0, stream, hist, pl.nchunk, totals);
    else
      hipLaunchKernelGGL(rs_scan_kernel, dim3((bins + RS_NW - 1) / RS_NW), dim3(RS_T), 0, stream, hist, pl.nchunk, bins, totals);
    size_t lds = (size_t)(1 + RS_NW) * bins * sizeof(unsigned int);
    if (pl.stage) lds += (size_t)(bins + (bins & 1)) * sizeof(unsigned int) + (size_t)RS_CHUNK * 12;
#define GR_RS_SCATTER(IO, ST)                                                                                              \
  hipLaunchKernelGGL((rs_scatter_kernel<IO, ST>), dim3(pl.nchunk), dim3(RS_T), lds, stream, kin, vin, iota ? iota_period : (int64_t)1, \
                     n, shift, nb, bins, hist, totals, ko, vo)
    if (iota) {
      if (pl.stage) GR_RS_SCATTER(true, true); else GR_RS_SCATTER(true, false);
    } else {
      if (pl.stage) GR_RS_SCATTER(false, true); else GR_RS_SCATTER(false, false);
    }
#undef GR_RS_SCATTER
    GR_LAUNCH_CHECK();
    kin = ko;
    vin = vo;
  }
  return GR_OK;
}

}  // namespace

size_t sort_pairs_temp_bytes(int64_t n) {
  if (n <= 0) return 256;
  const int64_t nchunk = (n + RS_CHUNK - 1) / RS_CHUNK;
  Carver cv(nullptr);
  cv.take<uint64_t>(n);
  cv.take<int32_t>(n);
  cv.take<uint32_t>((size_t)nchunk * RS_MAX_BINS);
  cv.take<uint32_t>(RS_MAX_BINS);
  return cv.used() + 256;
}

int sort_pairs_u64_i32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                       const int32_t* vals_in, int32_t* vals_out, int64_t n, int begin_bit,
                       int end_bit, hipStream_t stream) {
  if (n <= 0) return GR_OK;
  if (end_bit <= begin_bit) {  // nothing to sort on: stable == copy
    GR_HIP(hipMemcpyAsync(keys_out, keys_in, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, stream));
    GR_HIP(hipMemcpyAsync(vals_out, vals_in, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, stream));
    return GR_OK;
  }
  return radix_sort(temp, temp_bytes, keys_in, keys_out, vals_in, 0, vals_out, n, begin_bit, end_bit, stream);
}

// Same sort with the value stream generated on the fly: value(i) = i mod period -- nobody has to write or read a
// 4 B/entry iota array.
int sort_pairs_u64_iota(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, int64_t period,
                        int32_t* vals_out, int64_t n, int begin_bit, int end_bit, hipStream_t stream) {
  if (n <= 0) return GR_OK;
  GR_REQUIRE(end_bit > begin_bit && period > 0, "sort_pairs_u64_iota: empty bit range or period");
  return radix_sort(temp, temp_bytes, keys_in, keys_out, nullptr, period, vals_out, n, begin_bit, end_bit, stream);
}

}  // namespace gr
