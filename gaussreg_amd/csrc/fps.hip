// Farthest point sampling in stack mode ("next" row).
// Stands in for fpsample.bucket_fps_kdline_sampling(points, n, h=9) that GaussReg calls before collation
// (experiments/.../demo.py:46, test.py:46, datasets/.../dataset.py:127; pin environment.yaml:66).
// PARITY UNPINNED: fpsample (Rust) is not in the reference tree; it computes exact FPS with a kd-bucket
// acceleration, so from the same start index the sample SET is the one below up to ties; the start index
// is the caller's (the package draws it at random).  Contract here: sample 0 = start_idx; sample k+1 =
// arg max_i min_{j<=k} |p_i - p_sample_j|^2 in fp32 ((dx*dx + dy*dy) + dz*dz), lowest index on ties.
//
// FPS is sequential in k, and a global arg-max costs an exchange between the workgroups of a cloud:
//   * a cloud is split over G workgroups (G*batch <= 256, so all of them are co-resident), each keeping its
//     points AND their running min-distance in registers (PPT <= 20 per thread);
//   * fps_multi_kernel (the path every call of this repo takes): one exchange per ROUND, and a round accepts every sample
//     the sequential algorithm would take from the round's candidate set -- ~90 at 30 000 of 200 000 (see below);
//   * fps_kernel<0> (slabs of more than 20 points per thread: distances streamed from L2, 4 loads in flight): one sample
//     per exchange -- a workgroup publishes its best candidate {d, idx, x, y, z} in a double-buffered slot, every word
//     tagged with the iteration number; lane g of wave 0 in every workgroup polls slot g until all five tags are current,
//     then the wave reduces the G candidates -- one store + one load round trip per iteration, no fences, no counters;
//   * the spin is bounded: a lane that waits > 2^22 polls raises an error flag and all leave (no GPU hang).
#include <vector>

#include <atomic>

#include "common.hpp"

namespace gr {
namespace {
// Test switch (gr_fps_debug_force_fallback): 0 = normal; 1 = the co-operative launch of attempt 0 counts as refused by the
// runtime; 2 = attempt 0 runs and its result is discarded as if the inter-workgroup exchange had timed out.  Either way the
// call must finish through the one-workgroup-per-cloud retry with the same indices.
std::atomic<int> g_fps_force{0};


constexpr int FPS_T = 1024;
constexpr int FPS_GMAX = 64;

struct FpsCand {  // 64 B: five (iteration tag << 32 | payload) words {d, idx, x, y, z}, agent-scope accesses
  unsigned long long w[8];
};

__device__ __forceinline__ unsigned long long fps_key(float d, int idx) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (0xffffffffu - (unsigned)idx);  // d >= 0: bits are ordered
}

template <int PPT>
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float* __restrict__ pts, const int32_t* __restrict__ off,
                                                    const int32_t* __restrict__ samp_off,
                                                    const int32_t* __restrict__ start_idx, float* __restrict__ mind,
                                                    FpsCand* __restrict__ cand,
                                                    int* __restrict__ err, int G, int64_t* __restrict__ out) {
  __shared__ unsigned long long s_key[FPS_T / WAVE];
  __shared__ float s_c[4];
  __shared__ int s_abort;
  const int b = blockIdx.x / G, part = blockIdx.x % G;
  const int p0 = off[b], n = off[b + 1] - p0;
  const int o0 = samp_off[b], k = samp_off[b + 1] - o0;
  if (n <= 0 || k <= 0) return;
  const float* P = pts + 3 * (int64_t)p0;
  float* D = mind + p0;
  // this workgroup's slab [lo, hi): whole multiples of FPS_T except the last
  const int per = ((n + G - 1) / G + FPS_T - 1) / FPS_T * FPS_T;
  const int lo = min(part * per, n), hi = min(lo + per, n);
  float px[PPT > 0 ? PPT : 1], py[PPT > 0 ? PPT : 1], pz[PPT > 0 ? PPT : 1], pd[PPT > 0 ? PPT : 1];
  if (PPT > 0) {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = lo + j * FPS_T + threadIdx.x;
      const bool v = i < hi;
      px[j] = v ? P[3 * i] : 0.f;
      py[j] = v ? P[3 * i + 1] : 0.f;
      pz[j] = v ? P[3 * i + 2] : 0.f;
      pd[j] = v ? INFINITY : -1.0f;  // -1: never beats a real candidate (all real d >= 0)
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += FPS_T) D[i] = INFINITY;
  }
  int cur = start_idx ? min(max(start_idx[b], 0), n - 1) : 0;
  float cx = P[3 * cur], cy = P[3 * cur + 1], cz = P[3 * cur + 2];
  if (part == 0 && threadIdx.x == 0) out[o0] = cur;
  if (threadIdx.x == 0) s_abort = 0;
  FpsCand* slots = cand + (size_t)b * 2 * FPS_GMAX;

  for (int s = 1; s < k; ++s) {
    float best = -1.0f;
    int bi = 0;
    if (PPT > 0) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
        const float d = fminf(pd[j], (dx * dx + dy * dy) + dz * dz);
        pd[j] = d;
        if (d > best) {  // ascending index within the thread: the first maximum is kept
          best = d;
          bi = lo + j * FPS_T + threadIdx.x;
        }
      }
    } else {
      int i = lo + threadIdx.x;
      for (; i + 3 * FPS_T < hi; i += 4 * FPS_T) {
        float x[4], y[4], z[4], dd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = i + u * FPS_T;
          x[u] = P[3 * q], y[u] = P[3 * q + 1], z[u] = P[3 * q + 2], dd[u] = D[q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float dx = x[u] - cx, dy = y[u] - cy, dz = z[u] - cz;
          const float d = fminf(dd[u], (dx * dx + dy * dy) + dz * dz);
          D[i + u * FPS_T] = d;
          if (d > best) best = d, bi = i + u * FPS_T;
        }
      }
      for (; i < hi; i += FPS_T) {
        const float dx = P[3 * i] - cx, dy = P[3 * i + 1] - cy, dz = P[3 * i + 2] - cz;
        const float d = fminf(D[i], (dx * dx + dy * dy) + dz * dz);
        D[i] = d;
        if (d > best) best = d, bi = i;
      }
    }
    unsigned long long key = best >= 0.f ? fps_key(best, bi) : 0ull;
#pragma unroll
    for (int dd = WAVE / 2; dd > 0; dd >>= 1) {
      const unsigned long long o = __shfl_xor(key, dd, WAVE);
      key = o > key ? o : key;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) s_key[threadIdx.x / WAVE] = key;
    __syncthreads();
    if (threadIdx.x < WAVE) {
      unsigned long long v = threadIdx.x < FPS_T / WAVE ? s_key[threadIdx.x] : 0ull;
#pragma unroll
      for (int dd = WAVE / 2; dd > 0; dd >>= 1) {
        const unsigned long long o = __shfl_xor(v, dd, WAVE);
        v = o > v ? o : v;
      }
      // v = this workgroup's best; fetch its coordinates (a uniform, cached load)
      int wi = v ? (int)(0xffffffffu - (unsigned)(v & 0xffffffffu)) : 0;
      float wx = P[3 * wi], wy = P[3 * wi + 1], wz = P[3 * wi + 2];
      if (G > 1) {
        // tagged exchange: every 64-bit word carries the iteration number in its high half, so a reader knows a
        // word is current without any fence or counter (relaxed agent-scope accesses go past the per-XCD L2s)
        FpsCand* slot = slots + (size_t)(s & 1) * FPS_GMAX;
        const unsigned long long tag = (unsigned long long)(unsigned)s << 32;
        if (threadIdx.x == 0) {
          unsigned long long* w = reinterpret_cast<unsigned long long*>(&slot[part]);
          __hip_atomic_store(w + 0, tag | (unsigned)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(w + 1, tag | (unsigned)wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(w + 2, tag | __float_as_uint(wx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(w + 3, tag | __float_as_uint(wy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(w + 4, tag | __float_as_uint(wz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long ck = 0ull;
        unsigned ux = 0, uy = 0, uz = 0;
        bool bad = false;
        if ((int)threadIdx.x < G) {
          const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&slot[threadIdx.x]);
          int spins = 0;
          for (;;) {
            unsigned long long r[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) r[u] = __hip_atomic_load(w + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 5; ++u) ok = ok && (r[u] >> 32) == (unsigned)s;
            if (ok) {
              ck = ((r[0] & 0xffffffffull) << 32) | (0xffffffffu - (unsigned)r[1]);
              ux = (unsigned)r[2], uy = (unsigned)r[3], uz = (unsigned)r[4];
              break;
            }
            if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bad = true;
              break;
            }
          }
        }
        if (__any(bad)) {
          if (threadIdx.x == 0) s_abort = 1;
        }
#pragma unroll
        for (int dd = WAVE / 2; dd > 0; dd >>= 1) {
          const unsigned long long ok = __shfl_xor(ck, dd, WAVE);
          const unsigned ox = __shfl_xor(ux, dd, WAVE), oy = __shfl_xor(uy, dd, WAVE), oz = __shfl_xor(uz, dd, WAVE);
          if (ok > ck) ck = ok, ux = ox, uy = oy, uz = oz;
        }
        wi = (int)(0xffffffffu - (unsigned)(ck & 0xffffffffu));
        wx = __uint_as_float(ux), wy = __uint_as_float(uy), wz = __uint_as_float(uz);
      }
      if (threadIdx.x == 0) {
        s_c[0] = wx, s_c[1] = wy, s_c[2] = wz, s_c[3] = __int_as_float(wi);
        if (part == 0) out[o0 + s] = wi;
      }
    }
    __syncthreads();
    cx = s_c[0], cy = s_c[1], cz = s_c[2];
    cur = __float_as_int(s_c[3]);
    if (G > 1 && s_abort) return;  // barrier timed out: the whole workgroup leaves (uniform: read after the sync)
  }
}


// ---------------------------------------------------------------- many samples per exchange round
// FPS is a chain of global arg-max steps, and one step costs a whole inter-workgroup exchange (~3 us).  But a round can run
// the chain on a small CANDIDATE SET exactly: let E be the published keys (per-thread bests that survived the wave / workgroup
// selections) above B, where B bounds every key that is not in E (threads' runner-ups, what waves and workgroups held back).
// Points outside E only ever lose distance, so as long as the largest CURRENT key inside E beats B it is the global arg-max:
//   repeat { c = arg max of the current keys in E; stop if key(c) <= B; accept c; d(e) = min(d(e), |e - c|^2) for e in E }
// is the sequential algorithm, restricted to E.  Nearly all of E is isolated (no other candidate within sqrt(d) of it, in
// either direction): those keep their key, are accepted outright and only have to be RANKED by key for the output order
// (accepted keys are strictly decreasing along the exact sequence).  The sequential loop runs over the few candidates that
// are in some conflict.  At 30 000 of 200 000 points a round accepts ~90 samples (tools/fps_round_model.py: 338 rounds;
// the round-2 rule -- sorted top-32, longest conflict-free prefix -- needed 1 299).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int FPS_KPL = 8;                     // keys of one slot a polling lane holds (sorted, largest first)
constexpr int FPS_MW = 4;                      // candidates a single wave passes up (the rest raises the bound B)
constexpr int FPS_EC = 256;                    // capacity of the candidate set E (B is raised to the largest 5th key if needed)
constexpr int FPS_SLOT_STRIDE = 40;            // words between slots.  A workgroup publishes M = 8 keys + its bound (one 128-byte
                                               // line) or, when few workgroups share a cloud, M = 16 / 32 + bound (polled by two /
                                               // four lanes, 64 bytes each); every word carries a 1-bit tag in bit 63
                                               // (keys use 63 bits: d >= 0 has a clear sign bit)

// Wave-wide maximum through the DPP lanes-shift network (row_shr 1/2/4/8, row_bcast 15/31): six dependent VALU ops
// instead of six ds_bpermute round trips (~0.4 us each way for a 64-bit butterfly).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8 -> lane 15 of a row
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// A single wave runs these reductions back to back (about 7 cycles per dependent instruction), so the instruction count
// is the cost: the low half only needs its own reduction when two lanes tie on the high half (distances equal to the bit).
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_max_u32(hi);
  const unsigned long long tie = __ballot(hi == mh);  // never empty
  unsigned ml;
  if ((tie & (tie - 1ull)) == 0ull) ml = (unsigned)__builtin_amdgcn_readlane((int)lo, (int)__builtin_ctzll(tie));
  else ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

// The sequential part of a round, by ONE wave, over the nq candidates that are in some conflict (s_q lists their positions in
// E).  Every step accepts the largest current key while it beats `bound` and lowers the others; what is left at the end is
// rejected (s_fkey = 0: the points stay in the cloud with the distances the next fold gives them).  Returns the number of
// rejected candidates.  This form keeps one candidate per lane in registers (nq <= 64, nearly every round).
__device__ __forceinline__ int fps_resolve_conflicts(int nq, unsigned long long bound, const int* __restrict__ s_q,
                                                     const float4* __restrict__ s_cand,
                                                     const unsigned long long* __restrict__ s_ekey,
                                                     unsigned long long* __restrict__ s_fkey, int lane) {
  bool live = lane < nq;
  const int qi = s_q[live ? lane : 0];
  const float4 c = s_cand[qi];
  float cur = c.w;
  const unsigned lo = (unsigned)s_ekey[qi];
  for (int step = 0; step < nq; ++step) {
    const unsigned long long mine = live ? ((unsigned long long)__float_as_uint(cur) << 32) | lo : 0ull;
    const unsigned long long w = wave_max_u64(mine);
    if (w <= bound) break;  // also when nothing is live (w = 0)
    const bool own = mine == w;  // keys are unique: one lane
    if (own) {
      live = false;
      s_fkey[qi] = w;  // retired with its final key
    }
    const int ol = (int)__builtin_ctzll(__ballot(own));
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.x), ol));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.y), ol));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.z), ol));
    const float dx = c.x - wx, dy = c.y - wy, dz = c.z - wz;
    cur = fminf(cur, (dx * dx + dy * dy) + dz * dz);  // the fold's own expression: the same bits
  }
  if (live) s_fkey[qi] = 0ull;
  return (int)__builtin_popcountll(__ballot(live));
}

// The same for any nq (up to the capacity of E): current distances live in s_cur, a retired entry has s_cur < 0.
__device__ __forceinline__ int fps_resolve_conflicts_lds(int nq, unsigned long long bound, const int* __restrict__ s_q,
                                                         const float4* __restrict__ s_cand,
                                                         const unsigned long long* __restrict__ s_ekey,
                                                         unsigned long long* __restrict__ s_fkey, float* __restrict__ s_cur,
                                                         int lane) {
  for (int t = lane; t < nq; t += WAVE) s_cur[t] = s_cand[s_q[t]].w;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int np = (nq + WAVE - 1) / WAVE * WAVE;
  int accepted = 0;
  for (int step = 0; step < nq; ++step) {
    unsigned long long mine = 0ull;
    int mt = 0;
    for (int t = lane; t < nq; t += WAVE) {
      const float cu = s_cur[t];
      const unsigned long long kk = cu >= 0.f ? ((unsigned long long)__float_as_uint(cu) << 32) | (unsigned)s_ekey[s_q[t]] : 0ull;
      if (kk > mine) mine = kk, mt = t;
    }
    const unsigned long long w = wave_max_u64(mine);
    if (w <= bound) break;
    const bool own = mine == w;
    const int wt = __builtin_amdgcn_readlane(mt, (int)__builtin_ctzll(__ballot(own)));
    const int wq = s_q[wt];
    const float4 wc = s_cand[wq];
    if (lane == 0) s_fkey[wq] = w;
    ++accepted;
    for (int t = lane; t < np; t += WAVE) {
      if (t < nq) {
        const float4 c = s_cand[s_q[t]];
        const float dx = c.x - wc.x, dy = c.y - wc.y, dz = c.z - wc.z;
        const float cu = s_cur[t];
        s_cur[t] = t == wt ? -1.0f : (cu >= 0.f ? fminf(cu, (dx * dx + dy * dy) + dz * dz) : cu);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int t = lane; t < nq; t += WAVE)
    if (s_cur[t] >= 0.f) s_fkey[s_q[t]] = 0ull;
  return nq - accepted;
}

// PPT = points per thread held in registers; instantiated for 4 / 7 / 10 / 13 / 16 / 20 so that a slab pays for the
// points it has (a 12.2-point slab in the 20-point variant folded 64 % more distances than it owned).  Smaller
// workgroups (256 threads, four per CU, to overlap one workgroup's exchange with the others' folding) measured SLOWER:
// the exchange gets four times the slots to poll.
//
// Bucket pruning (what fpsample's kd-line buckets do on the CPU, here at wave granularity): the cloud arrives sorted along a
// Hilbert curve (`spts`, with `perm` = position -> original index), every wave owns a CONTIGUOUS run of it and keeps that
// run's bounding box and the largest running distance of its points.  A new sample that is farther from the box than
// that distance cannot lower any of them: the wave skips the fold, and a wave no sample reached this round also keeps
// the candidates it published last round.  After the first few hundred samples that is almost every wave in almost
// every round; results are exactly those of the unpruned algorithm (keys carry the ORIGINAL index).
//
// MW = keys a wave passes up.  (Measured and taken out: a "wide" variant -- 32 keys per workgroup, two points offered per
// thread -- and a Morton curve order)
template <int PPT, int M, int MW = FPS_MW>
__global__ __launch_bounds__(FPS_T) void fps_multi_kernel(const float* __restrict__ pts, const float* __restrict__ spts,
                                                          const int32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ off,
                                                          const int32_t* __restrict__ samp_off,
                                                          const int32_t* __restrict__ start_idx,
                                                          unsigned long long* __restrict__ slots_all,
                                                          int* __restrict__ err, int G, int64_t* __restrict__ out) {
  constexpr int KPL = FPS_KPL, LPS = M / KPL, NW = FPS_T / WAVE;  // LPS: polling lanes per slot
  constexpr int EC = M > 16 ? 512 : FPS_EC;                       // M = 32 runs with <= 16 workgroups: E never exceeds 512
  constexpr int NWK = NW * MW, KL = (NWK + WAVE - 1) / WAVE;      // wave-level candidates of the workgroup; per lane of wave 0
  static_assert(M % KPL == 0 && FPS_SLOT_STRIDE > M && NWK % 2 == 0 && M < NWK, "layout");
  constexpr bool ROWS = PPT >= 13;  // pays where a wave's run is long; the 20-point variant has no registers left for it (+11 spills: +1 %)
  constexpr int UN = PPT <= 10 ? 4 : 2;  // partners in flight in stages 4 and 6: the large slabs have no registers to spare
  extern __shared__ int s_perm[];  // [PPT][FPS_T] original (cloud-local) index of every point this workgroup holds
  __shared__ __attribute__((aligned(16))) unsigned long long s_wtop[NWK];
  __shared__ unsigned long long s_sel[KL * WAVE];                 // the same keys in descending order
  __shared__ unsigned long long s_wbound[NW];
  __shared__ __attribute__((aligned(16))) float4 s_acc[EC];   // the samples accepted in the last round
  __shared__ __attribute__((aligned(16))) float4 s_cand[EC + WAVE];  // E: {x, y, z, d}; padded for stage 4
  __shared__ unsigned long long s_ekey[EC + WAVE], s_fkey[EC];   // key on entry / when accepted (0: rejected)
  __shared__ int s_rank[EC], s_flag[EC], s_q[EC];
  __shared__ float s_cur[EC];
  __shared__ unsigned long long s_qe[EC], s_qf[EC];     // the conflict candidates' keys on entry / when accepted
  __shared__ int s_na, s_abort, s_c, s_nq, s_nqc;
  __shared__ unsigned long long s_bound;
  const int b = blockIdx.x / G, part = blockIdx.x % G;
  const int p0 = off[b], n = off[b + 1] - p0;
  const int o0 = samp_off[b], k = samp_off[b + 1] - o0;
  if (n <= 0 || k <= 0) return;
  const float* P = pts + 3 * (int64_t)p0;    // original order: candidate coordinates by original index
  const float* S = spts + 3 * (int64_t)p0;   // curve order: the points this thread folds
  const int per = ((n + G - 1) / G + FPS_T - 1) / FPS_T * FPS_T;
  const int lo = min(part * per, n), hi = min(lo + per, n);
  const int lane = threadIdx.x & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x / WAVE);  // scalar: a wave's index
  const int ppt_used = per / FPS_T;              // <= PPT
  const int wave_lo = lo + wv * ppt_used * WAVE;  // this wave's run: ppt_used * 64 consecutive curve positions
  float px[PPT], py[PPT], pz[PPT], pd[PPT];
  float bx0 = INFINITY, by0 = INFINITY, bz0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY, bz1 = -INFINITY;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = wave_lo + j * WAVE + lane;
    const bool v = j < ppt_used && i < hi;
    px[j] = v ? S[3 * i] : 0.f;
    py[j] = v ? S[3 * i + 1] : 0.f;
    pz[j] = v ? S[3 * i + 2] : 0.f;
    pd[j] = v ? INFINITY : -1.0f;  // an empty slot never becomes a candidate (the fold's min keeps the -1)
    s_perm[j * FPS_T + threadIdx.x] = v ? perm[p0 + i] : 0;
    if (v) {
      bx0 = fminf(bx0, px[j]), by0 = fminf(by0, py[j]), bz0 = fminf(bz0, pz[j]);
      bx1 = fmaxf(bx1, px[j]), by1 = fmaxf(by1, py[j]), bz1 = fmaxf(bz1, pz[j]);
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    bx0 = fminf(bx0, __shfl_xor(bx0, d, WAVE)), by0 = fminf(by0, __shfl_xor(by0, d, WAVE)), bz0 = fminf(bz0, __shfl_xor(bz0, d, WAVE));
    bx1 = fmaxf(bx1, __shfl_xor(bx1, d, WAVE)), by1 = fmaxf(by1, __shfl_xor(by1, d, WAVE)), bz1 = fmaxf(bz1, __shfl_xor(bz1, d, WAVE));
  }
  // wave-uniform from here on: keep the box in scalar registers (six vector registers matter at 20 points per thread)
  bx0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx0)));
  by0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(by0)));
  bz0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bz0)));
  bx1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx1)));
  by1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(by1)));
  bz1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bz1)));
  // Second level of the pruning: lane r < NP keeps the box of ROW PAIR r (points 2r and 2r + 1 of every lane: 128
  // consecutive curve positions, the unit of the packed fold).  A sample that reaches the wave's box is tested against
  // the row pairs before it is folded: the busiest waves are those with sparse or stretched runs, where a sample passes
  // the wave test but lowers one or two rows (tools/fps_round_model.py --fold-stats: 20 % of the row pairs, and the
  // critical path of the folds -- the busiest wave of every round -- shrinks by 47 %).
  constexpr int NP = (PPT + 1) / 2;
  __shared__ float s_rbox[ROWS ? NW * 6 * 16 : 1];  // [wave][6][row pair]: read back at the start of every round (kept in
                                                    // registers across the round they cost the 20-point variant 30 spills)
#pragma unroll
  for (int r = 0; ROWS && r < NP; ++r) {
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 2 * r + u;
      if (j < PPT && pd[j < PPT ? j : 0] >= 0.f) {
        lo3[0] = fminf(lo3[0], px[j < PPT ? j : 0]), lo3[1] = fminf(lo3[1], py[j < PPT ? j : 0]), lo3[2] = fminf(lo3[2], pz[j < PPT ? j : 0]);
        hi3[0] = fmaxf(hi3[0], px[j < PPT ? j : 0]), hi3[1] = fmaxf(hi3[1], py[j < PPT ? j : 0]), hi3[2] = fmaxf(hi3[2], pz[j < PPT ? j : 0]);
      }
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        lo3[c] = fminf(lo3[c], __shfl_xor(lo3[c], d, WAVE));
        hi3[c] = fmaxf(hi3[c], __shfl_xor(hi3[c], d, WAVE));
      }
    }
    if (lane < 3) s_rbox[(wv * 6 + lane) * 16 + r] = lane == 0 ? lo3[0] : lane == 1 ? lo3[1] : lo3[2];
    else if (lane < 6) s_rbox[(wv * 6 + lane) * 16 + r] = lane == 3 ? hi3[0] : lane == 4 ? hi3[1] : hi3[2];
  }
  float wave_maxd = INFINITY;  // largest running distance among this wave's points (wave-uniform)
  bool owner = false;          // this lane's best point is one of the wave's published candidates ...
  int owner_bdi = 0;           // ... and had these distance bits when it was selected
  const int start = start_idx ? min(max(start_idx[b], 0), n - 1) : 0;
  if (threadIdx.x == 0) {
    s_acc[0] = make_float4(P[3 * start], P[3 * start + 1], P[3 * start + 2], 0.f);
    s_na = 1;
    s_abort = 0;
    if (part == 0) out[o0] = start;
  }
  __syncthreads();
  unsigned long long* slots = slots_all + (size_t)b * 2 * FPS_GMAX * FPS_SLOT_STRIDE;
  int count = 1;
  // rounds start at 2: buffer (round & 1) is reused every second round and its words carry the tag bit
  // (round >> 1) & 1, which flips between consecutive uses; zero-initialised slots read as tag 0, first uses expect 1
  for (unsigned round = 2; count < k; ++round) {
    // Values derived from the thread / lane index (LDS addresses, masks) are recomputed every round: hoisted out of the loop
    // they were spilled, and a scratch reload costs more than the one or two instructions that rebuild them.
    int tid = (int)threadIdx.x, ln = lane;
    if (PPT > 10) asm volatile("" : "+v"(tid), "+v"(ln));  // (the small variants have registers to spare)
    // ---- 1. fold the samples accepted last round into the running distances; per-thread best and runner-up
    const int na = s_na;
    if (tid < EC) s_rank[tid] = 0, s_flag[tid] = 0;  // for stage 4 (read after two barriers)
    if (tid < KL * WAVE) s_sel[tid] = 0ull;          // for stage 3a (last read in stage 3 of the previous round)
    if (tid == 0) s_nqc = 0;                         // stage 4's list of the candidates in conflict
    bool touched = round == 2;
    float rb0x = 0.f, rb0y = 0.f, rb0z = 0.f, rb1x = 0.f, rb1y = 0.f, rb1z = 0.f;
    if (ROWS) {
      const float* rb = s_rbox + wv * 6 * 16 + min(ln, NP - 1);
      rb0x = rb[0], rb0y = rb[16], rb0z = rb[32], rb1x = rb[48], rb1y = rb[64], rb1z = rb[80];
    }
    for (int base = 0; base < na; base += WAVE) {
      // lanes = samples: distance from each new sample to the wave's box, rounded down by more than the fold's own
      // rounding (8 ulp-steps); at or beyond wave_maxd no running distance of this wave can drop.  The first round folds
      // everything (a wave without points has an empty box: inf >= inf would skip it and leave its slots unwritten).
      unsigned long long todo;
      {
        const float4 a4 = s_acc[min(base + ln, EC - 1)];
        const float ex = fmaxf(fmaxf(bx0 - a4.x, a4.x - bx1), 0.f), ey = fmaxf(fmaxf(by0 - a4.y, a4.y - by1), 0.f),
                    ez = fmaxf(fmaxf(bz0 - a4.z, a4.z - bz1), 0.f);
        const float lb = ((ex * ex + ey * ey) + ez * ez) * (1.0f - 9.5367431640625e-7f);
        todo = __ballot(base + ln < na && (round == 2 || !(lb >= wave_maxd)));
      }
      touched = touched || todo != 0ull;
      while (todo) {
        const int a = base + (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        // sample-major: one LDS read of the sample, then all of this thread's points two at a time (packed fp32 -- the
        // same sub / mul / add sequence as the scalar form, so distances stay bit-identical)
        const float4 a4 = s_acc[a];
        unsigned rows = 0xffffffffu;  // row pairs the sample can reach (same bound as the wave test, lanes = row pairs)
        if (ROWS) {
          const float ex = fmaxf(fmaxf(rb0x - a4.x, a4.x - rb1x), 0.f), ey = fmaxf(fmaxf(rb0y - a4.y, a4.y - rb1y), 0.f),
                      ez = fmaxf(fmaxf(rb0z - a4.z, a4.z - rb1z), 0.f);
          const float lb = ((ex * ex + ey * ey) + ez * ez) * (1.0f - 9.5367431640625e-7f);
          rows = (unsigned)__ballot(ln < NP && (round == 2 || !(lb >= wave_maxd)));
        }
        const f32x2 ax2 = {a4.x, a4.x}, ay2 = {a4.y, a4.y}, az2 = {a4.z, a4.z};
#pragma unroll
        for (int j = 0; j + 1 < PPT; j += 2) {
          if ((rows >> (j / 2)) & 1u) {  // wave-uniform
            const f32x2 dx = f32x2{px[j], px[j + 1]} - ax2, dy = f32x2{py[j], py[j + 1]} - ay2, dz = f32x2{pz[j], pz[j + 1]} - az2;
            const f32x2 dd = (dx * dx + dy * dy) + dz * dz;
            pd[j] = fminf(pd[j], dd.x);
            pd[j + 1] = fminf(pd[j + 1], dd.y);
          }
        }
        if ((PPT & 1) && ((rows >> (PPT / 2)) & 1u)) {
          const float dx = px[PPT - 1] - a4.x, dy = py[PPT - 1] - a4.y, dz = pz[PPT - 1] - a4.z;
          pd[PPT - 1] = fminf(pd[PPT - 1], (dx * dx + dy * dy) + dz * dz);
        }
      }
    }
    if (touched) {  // wave-uniform; otherwise s_wtop / s_wbound still hold this wave's candidates of the last round
      // the thread's best key and a bound for its runner-up.  Distances are >= 0 (empty slots hold -1), so their bit
      // patterns order like the values: the largest distance by integer maxima, then one pass that counts its occurrences,
      // keeps the largest OTHER distance and the slot -- five instructions per point, ONE index read from LDS (building
      // all the 64-bit keys was 13 instructions and a read per point).  The runner-up only feeds the bound B, so
      // (distance, lowest index) bounds its key from above.  Equal distances inside a thread: the exact keys decide.
      int bdi = __float_as_int(pd[0]);
#pragma unroll
      for (int j = 1; j < PPT; ++j) bdi = max(bdi, __float_as_int(pd[j]));
      // The wave's published keys belong to `owner` lanes.  If none of THEIR best distances moved, the top-MW keys are what
      // they were (a key = distance + index of the same point) and the old bound still bounds (distances only fall): the
      // samples of this round lowered points the wave never offered, and the selection below can be skipped.
      if (round > 2 && !__any(owner && bdi != owner_bdi)) touched = false;
    }
    if (touched) {
      int bdi = __float_as_int(pd[0]);
#pragma unroll
      for (int j = 1; j < PPT; ++j) bdi = max(bdi, __float_as_int(pd[j]));
      const int none = __float_as_int(-1.0f);
      int cnt = 0, bj = 0, sdi = none;
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int dv = __float_as_int(pd[j]);
        const bool e = dv == bdi;
        cnt += e ? 1 : 0;
        sdi = max(sdi, e ? none : dv);
        bj = e ? j : bj;
      }
      unsigned long long best = 0ull, second = 0ull;
      if (bdi >= 0) {
        best = ((unsigned long long)(unsigned)bdi << 32) | (0xffffffffu - (unsigned)s_perm[bj * FPS_T + tid]);
        second = sdi >= 0 ? ((unsigned long long)(unsigned)sdi << 32) | 0xffffffffull : 0ull;
      }
      if (cnt > 1 && bdi >= 0) {
        best = 0ull, second = 0ull;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
          if (__float_as_int(pd[j]) >= 0) {
            const unsigned long long kk = fps_key(pd[j], s_perm[j * FPS_T + tid]);
            if (kk > best) {
              second = best;
              best = kk;
            } else if (kk > second) {
              second = kk;
            }
          }
        }
      }
      // ---- 2. top-MW of the wave's bests (unique keys: exactly one lane owns each maximum); the rest bounds B
      unsigned long long mine = best;
      owner = false;
      owner_bdi = bdi;
#pragma unroll
      for (int r = 0; r < MW; ++r) {
        const unsigned long long w = wave_max_u64(mine);
        if (ln == 0) s_wtop[wv * MW + r] = w;
        if (r == 0) wave_maxd = __uint_as_float((unsigned)(w >> 32));  // keys order by distance first
        if (mine == w && w != 0ull) {
          mine = 0ull;
          owner = true;
        }
      }
      {
        const unsigned long long wb = wave_max_u64(mine > second ? mine : second);
        if (ln == 0) s_wbound[wv] = wb;
      }
    }
    __syncthreads();
    // ---- 3a. all waves: the workgroup's NWK wave-level candidates in descending order, ranked by counting (keys of points are
    // unique; empty slots are 0).  Wave w ranks its own MW keys, 64 / MW lanes per key, each lane against its share of the
    // NWK keys; the partial counts meet through lane shuffles.  (One wave ranking all of them -- one key per lane, NWK
    // compares each -- was 1.1 us of every round with the other fifteen waves waiting at the barrier.)
    {
      constexpr int LPK = WAVE / MW, TPL = NWK / LPK;  // lanes per key, keys a lane compares against
      static_assert(WAVE % MW == 0 && NWK % LPK == 0 && TPL % 2 == 0, "ranking layout");
      const unsigned long long mine = s_wtop[wv * MW + ln / LPK];
      const ulonglong2* w2 = reinterpret_cast<const ulonglong2*>(s_wtop) + (ln % LPK) * (TPL / 2);
      int rk = 0;
#pragma unroll
      for (int t = 0; t < TPL / 2; ++t) {
        const ulonglong2 q = w2[t];
        rk += (int)(q.x > mine) + (int)(q.y > mine);
      }
#pragma unroll
      for (int d = LPK / 2; d > 0; d >>= 1) rk += __shfl_xor(rk, d, WAVE);
      if (ln % LPK == 0 && mine != 0ull) s_sel[rk] = mine;
    }
    __syncthreads();
    // ---- 3. wave 0: workgroup top-M, exchange, the candidate set E
    if (wv == 0) {
      static_assert(KPL >= EC / WAVE, "E holds at most EC / 64 keys per polling lane once B is raised");
      static_assert(M < WAVE, "s_sel[M] is the largest key a workgroup holds back");
      const unsigned long long mykey = ln < M ? s_sel[ln] : 0ull;  // lane r < M: the r-th largest key
      unsigned long long bnd = wave_max_u64(ln < NW ? s_wbound[ln] : 0ull);
      {
        const unsigned long long rest = s_sel[M];
        bnd = rest > bnd ? rest : bnd;
      }
      bool bad = false;
      unsigned long long kk[KPL];  // lane g * LPS + h: keys [8 h, 8 h + 8) of workgroup g, largest first
      unsigned long long sb = bnd;
#pragma unroll
      for (int r = 0; r < KPL; ++r) kk[r] = 0ull;
      if (G > 1) {
        unsigned long long* buf = slots + (size_t)(round & 1u) * FPS_GMAX * FPS_SLOT_STRIDE;
        const unsigned long long tag = (unsigned long long)((round >> 1) & 1u) << 63;
        if (ln < M)
          __hip_atomic_store(buf + part * FPS_SLOT_STRIDE + ln, tag | mykey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (ln == M)
          __hip_atomic_store(buf + part * FPS_SLOT_STRIDE + M, tag | bnd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // lane g * LPS + h polls its eight keys of workgroup g's slot (and the bound) until they carry this round's tag bit
        sb = 0ull;
        if (ln < G * LPS) {
          const unsigned long long* w = buf + (ln / LPS) * FPS_SLOT_STRIDE;
          const int h = ln % LPS;
          int spins = 0;
          for (;;) {
            unsigned long long rd[KPL + 1];
#pragma unroll
            for (int u = 0; u < KPL; ++u) rd[u] = __hip_atomic_load(w + h * KPL + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rd[KPL] = __hip_atomic_load(w + M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < KPL + 1; ++u) ok = ok && ((rd[u] ^ tag) >> 63) == 0ull;
            if (ok) {
#pragma unroll
              for (int r = 0; r < KPL; ++r) kk[r] = rd[r] & ~(1ull << 63);
              sb = rd[KPL] & ~(1ull << 63);
              break;
            }
            if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bad = true;
              break;
            }
          }
        }
      } else {
        kk[0] = mykey;  // a single workgroup: one key per lane (lanes >= M hold 0)
      }
      // every point already at distance zero (more samples asked for than the cloud has distinct points): the arg-max is the
      // lowest index, and stays it -- the sequential algorithm returns index 0 from here on
      if ((wave_max_u64(kk[0]) >> 32) == 0ull && !__any(bad)) {
        if (ln == 0) s_c = -1;
      } else {
      // B: nothing outside the published keys exceeds the largest workgroup bound
      unsigned long long bound = wave_max_u64(sb);
      int cnt = 0;
#pragma unroll
      for (int r = 0; r < KPL; ++r) cnt += kk[r] > bound ? 1 : 0;  // sorted lists: a prefix
      int incl = wave_incl_scan_add_dpp(cnt);
      int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
      if (EC / WAVE < KPL && total > EC) {  // keep at most EC / 64 keys per polling lane: the largest dropped key joins the bound
        const unsigned long long cut = wave_max_u64(kk[EC / WAVE < KPL ? EC / WAVE : 0]);
        bound = cut > bound ? cut : bound;
        cnt = 0;
#pragma unroll
        for (int r = 0; r < KPL; ++r) cnt += kk[r] > bound ? 1 : 0;
        incl = wave_incl_scan_add_dpp(cnt);
        total = __builtin_amdgcn_readlane(incl, WAVE - 1);
      }
      {
        const int at = incl - cnt;
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
          if (r < cnt) s_ekey[at + r] = kk[r];
        }
      }
      if (__any(bad)) {
        if (ln == 0) s_abort = 1;
      }
      if (ln == 0) s_c = total, s_bound = bound;
      }
    }
    __syncthreads();
    if (s_abort) return;
    if (s_c < 0) {
      if (part == 0)
        for (int i = count + tid; i < k; i += FPS_T) out[o0 + i] = 0;
      return;
    }
    // ---- 4. all waves: which candidates are in a conflict (some other candidate within sqrt(d) of either of the two), and
    // every candidate's rank among the keys on entry.  Thread = (candidate i, share of the partners j); shares are
    // multiples of four partners, the tail of E is padded with points at infinity (no conflict, key 0).
    const int C = s_c;
    const int cshift = C <= 64 ? 6 : C <= 128 ? 7 : C <= 256 ? 8 : 9;  // candidates padded to a power of two, FPS_T >> cshift partner shares
    const int cpad = 1 << cshift;
    const int share = (((C + (FPS_T >> cshift) - 1) >> (10 - cshift)) + 3) & ~3;
    if (tid < C) {  // coordinates by original index (the cloud is read-only: plain cached loads)
      const unsigned long long ek = s_ekey[tid];
      const int ci = (int)(0xffffffffu - (unsigned)(ek & 0xffffffffull));
      s_cand[tid] = make_float4(P[3 * ci], P[3 * ci + 1], P[3 * ci + 2], __uint_as_float((unsigned)(ek >> 32)));
      s_fkey[tid] = ek;
    } else if (tid < share * (FPS_T >> cshift)) {  // <= C + 4 * 16
      s_cand[tid] = make_float4(INFINITY, 0.f, 0.f, 0.f);
      s_ekey[tid] = 0ull;
    }
    __syncthreads();
    const int ci_ = tid & (cpad - 1), cg = tid >> cshift;
    if (C > 1 && ci_ < C) {
      const int j0 = cg * share, j1 = j0 + share;
      const float4 me = s_cand[ci_];
      const unsigned long long mk = s_ekey[ci_];
      // the candidate meets itself in one share: |e - e|^2 = 0 < d counts unless d = 0
      int near = ci_ >= j0 && ci_ < j1 && me.w > 0.f ? -1 : 0, above = 0;
      for (int j = j0; j < j1; j += UN) {  // loads first, no short circuits (a conditional load costs an LDS round trip)
        float4 o[UN];
        unsigned long long ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) o[u] = s_cand[j + u], ok[u] = s_ekey[j + u];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const float dx = me.x - o[u].x, dy = me.y - o[u].y, dz = me.z - o[u].z;
          const float dd = (dx * dx + dy * dy) + dz * dz;
          // distances are >= 0 (or +inf for the padding): their bit patterns order like the values
          near += (int)(__float_as_uint(dd) < max(__float_as_uint(me.w), __float_as_uint(o[u].w)));
          above += (int)(ok[u] > mk);
        }
      }
      // the first share that sees a conflict of candidate ci_ appends it to the list of stage 5 (any order: the chain goes by
      // keys) -- wave 0 collecting the flags with ballots afterwards was 0.3 us of its serial section
      if (near > 0 && atomicExch(&s_flag[ci_], 1) == 0) s_q[atomicAdd(&s_nqc, 1)] = ci_;
      if (above) atomicAdd(&s_rank[ci_], above);
    }
    __syncthreads();
    // ---- 5. wave 0: the exact chain over the candidates in conflict
    if (wv == 0) {
      const int nq = s_nqc;
      const unsigned long long bound = s_bound;
      int rejected = 0;
      if (nq > WAVE) rejected = fps_resolve_conflicts_lds(nq, bound, s_q, s_cand, s_ekey, s_fkey, s_cur, ln);
      else if (nq > 0) rejected = fps_resolve_conflicts(nq, bound, s_q, s_cand, s_ekey, s_fkey, ln);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int t = ln; t < ((nq + 3) & ~3); t += WAVE) {  // keys of the conflict candidates, padded with zeros to four
        const int q = s_q[t < nq ? t : 0];
        s_qe[t] = t < nq ? s_ekey[q] : 0ull;
        s_qf[t] = t < nq ? s_fkey[q] : 0ull;
      }
      if (ln == 0) s_nq = nq, s_na = min(C - rejected, k - count);
    }
    __syncthreads();
    // ---- 6. all waves: output positions.  Accepted keys decrease strictly along the exact sequence, so a sample's
    // position is its rank among the accepted keys.  A candidate without conflict kept its key: its rank on entry minus
    // the conflict candidates that dropped below it.  A candidate in conflict is ranked from scratch (one wave each).
    {
      const int nq = s_nq, room = k - count;
      if (tid < C && s_flag[tid] == 0) {
        const unsigned long long mk = s_ekey[tid];
        int r = s_rank[tid];
        for (int u = 0; u < nq; u += UN) {  // (nq is padded to a multiple of four)
          unsigned long long qe[UN], qf[UN];
#pragma unroll
          for (int v = 0; v < UN; ++v) qe[v] = s_qe[u + v], qf[v] = s_qf[u + v];
#pragma unroll
          for (int v = 0; v < UN; ++v) r -= (int)(qe[v] > mk) & (int)(qf[v] < mk);
        }
        if (r < room) {
          s_acc[r] = s_cand[tid];
          if (part == 0) out[o0 + count + r] = (int)(0xffffffffu - (unsigned)(mk & 0xffffffffull));
        }
      }
      for (int u = wv; u < nq; u += NW) {
        const int q = s_q[u];
        const unsigned long long fk = s_fkey[q];
        if (fk == 0ull) continue;  // wave-uniform
        int r = 0;
        for (int j = ln; j < cpad; j += WAVE) r += (int)__builtin_popcountll(__ballot(j < C && s_fkey[j] > fk));
        if (ln == 0 && r < room) {
          const float4 c = s_cand[q];
          s_acc[r] = make_float4(c.x, c.y, c.z, 0.f);
          if (part == 0) out[o0 + count + r] = (int)(0xffffffffu - (unsigned)(fk & 0xffffffffull));
        }
      }
    }
    __syncthreads();
    count += s_na;
  }
}

// ---------------------------------------------------------------- space-filling-curve order (for the bucket pruning above)
__device__ __forceinline__ unsigned spread3(unsigned v) {  // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// Position of cell (x, y, z) of a 1024^3 grid along the 3-D Hilbert curve (Skilling's transpose form: undo the excess
// work of the Gray code bit plane by bit plane, Gray-encode, interleave).  Consecutive positions are ADJACENT cells, so a
// run of the sorted points is one connected blob; runs of a Morton curve regularly straddle one of its jumps and get a
// bounding box as large as the room -- which no sample misses (tools/fps_round_model.py --fold-stats: over a 200 k -> 30 k
// run with ten workgroups the busiest wave folds 18 991 samples in Morton order, 4 144 in Hilbert order).
__device__ __forceinline__ unsigned hilbert3_code(unsigned x0, unsigned x1, unsigned x2) {
  unsigned X[3] = {x0, x1, x2};
#pragma unroll
  for (unsigned Q = 512u; Q > 1u; Q >>= 1) {
    const unsigned P = Q - 1u;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) X[0] ^= P;
      else {
        const unsigned t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  unsigned t = 0u;
#pragma unroll
  for (unsigned Q = 512u; Q > 1u; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1u;
  return (spread3(X[0] ^ t) << 2) | (spread3(X[1] ^ t) << 1) | spread3(X[2] ^ t);
}

// key = cloud << 32 | 30-bit Hilbert code of the point inside its cloud's bounding cube (1024 cells per axis)
__global__ __launch_bounds__(256) void fps_curve_kernel(const float* __restrict__ pts, const int32_t* __restrict__ off, int nb,
                                                         const uint32_t* __restrict__ bbox, int n, int drop,
                                                         unsigned long long* __restrict__ keys, uint32_t* __restrict__ field) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = find_batch(off, nb, i);
  const float mnx = ord2f(bbox[b * 6]), mny = ord2f(bbox[b * 6 + 1]), mnz = ord2f(bbox[b * 6 + 2]);
  const float ext = fmaxf(fmaxf(ord2f(bbox[b * 6 + 3]) - mnx, ord2f(bbox[b * 6 + 4]) - mny), ord2f(bbox[b * 6 + 5]) - mnz);
  const float sc = ext > 0.f && isfinite(ext) ? 1023.0f / ext : 0.f;
  const unsigned cx = (unsigned)fminf(fmaxf((pts[3 * (int64_t)i] - mnx) * sc, 0.f), 1023.f);      // NaN -> 0
  const unsigned cy = (unsigned)fminf(fmaxf((pts[3 * (int64_t)i + 1] - mny) * sc, 0.f), 1023.f);
  const unsigned cz = (unsigned)fminf(fmaxf((pts[3 * (int64_t)i + 2] - mnz) * sc, 0.f), 1023.f);
  const unsigned code = hilbert3_code(cx, cy, cz);
  // (a prefix of a Hilbert code is the coarser curve)
  if (field) field[i] = (code >> drop) + 1u;  // the bucket sort's field (0 = "absent" there); the cloud is the segment
  else keys[i] = (((unsigned long long)(unsigned)b << 32) | code) >> drop;
}

// sorted position j holds global point vals[j]: copy its coordinates, keep its cloud-local index
__global__ __launch_bounds__(256) void fps_gather_kernel(const float* __restrict__ pts, const int32_t* __restrict__ off, int nb,
                                                         const int32_t* __restrict__ vals, int n, float* __restrict__ spts,
                                                         int32_t* __restrict__ perm) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int g = vals[j];
  spts[3 * (int64_t)j] = pts[3 * (int64_t)g];
  spts[3 * (int64_t)j + 1] = pts[3 * (int64_t)g + 1];
  spts[3 * (int64_t)j + 2] = pts[3 * (int64_t)g + 2];
  perm[j] = g - off[find_batch(off, nb, j)];  // clouds stay in place: position j and point g belong to the same cloud
}

}  // namespace
}  // namespace gr

namespace gr {
int g_fps_force_get() { return g_fps_force.load(); }
void g_fps_force_set(int m) { g_fps_force.store(m); }
}  // namespace gr

using namespace gr;

namespace gr {
namespace {
// Test switch (gr_fps_debug_bucket_sort): 0 = the curve pre-pass always takes the radix sort
std::atomic<int> g_fps_bucket_sort{1};
// the bucket sort of the curve pre-pass (depth_sort.hip, ragged segments = clouds): tables for up to this many (cloud, 2048-point
// chunk of the LONGEST cloud) rows -- a batch more ragged than that takes the radix sort
inline int64_t fps_ds_rows(int64_t n, int64_t batch) { return 2 * ((n + 2047) / 2048) + 2 * batch; }
inline size_t fps_ds_bytes(int64_t n, int64_t batch) {
  const size_t rows = (size_t)fps_ds_rows(n, batch);
  return align_up(rows * 512 * sizeof(uint16_t), 256) + align_up(rows * 512 * sizeof(uint32_t), 256) +
         align_up((size_t)batch * 512 * sizeof(int32_t), 256) + align_up((size_t)batch * 2 * sizeof(uint32_t), 256) +
         align_up(((size_t)batch * 512 + 1) * sizeof(int32_t), 256) + 512;
}
inline int fps_bits_for(unsigned long long v) {  // bits needed to represent values in [0, v)
  int b = 0;
  while (b < 64 && (1ull << b) < v) ++b;
  return b;
}
}  // namespace
}  // namespace gr

extern "C" int gr_fps_debug_bucket_sort(int on) {
  const int old = gr::g_fps_bucket_sort.load();
  if (on == 0 || on == 1) gr::g_fps_bucket_sort.store(on);
  return old;
}

extern "C" size_t gr_fps_workspace_bytes(int64_t n, int64_t batch) {
  if (n < 0 || batch < 0) return 0;
  return align_up(gr::fps_ds_bytes(n, batch), 256) + align_up((size_t)batch * 2 * 4, 256) + 2 * align_up((size_t)(batch + 1) * 4, 256) +
         align_up((size_t)n * 4, 256) + 3 * align_up((size_t)(batch + 1) * 4, 256) +
         align_up((size_t)batch * 2 * FPS_GMAX * sizeof(FpsCand), 256) + align_up((size_t)(batch + 1) * 4, 256) +
         align_up((size_t)batch * 2 * FPS_GMAX * FPS_SLOT_STRIDE * 8, 256) +
         // curve pre-pass: keys in/out, values out, sorted points, permutation, bounding boxes, sort scratch
         2 * align_up((size_t)n * 8, 256) + 2 * align_up((size_t)n * 4, 256) + align_up((size_t)n * 12, 256) +
         align_up((size_t)batch * 6 * 4, 256) + align_up((size_t)(batch + 1) * 4, 256) + align_up(sort_pairs_temp_bytes(n), 256) +
         512;
}

extern "C" int gr_fps(const float* points, const int64_t* h_lengths, const int64_t* h_num_samples,
                      const int64_t* h_start_indices, int64_t n, int64_t batch, int64_t* out_indices, void* ws,
                      size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && batch >= 0 && n < (1ll << 31) - 1 && batch < (1 << 20), "bad sizes");
  if (batch == 0 || n == 0) return GR_OK;
  GR_REQUIRE(points && h_lengths && h_num_samples && out_indices, "null argument");
  if (!ws || ws_bytes < gr_fps_workspace_bytes(n, batch)) {
    set_error("fps workspace too small");
    return GR_ERR_WORKSPACE;
  }
  std::vector<int32_t> off(batch + 1, 0), soff(batch + 1, 0), st(batch, 0);
  int64_t nmax = 0;
  for (int64_t b = 0; b < batch; ++b) {
    GR_REQUIRE(h_lengths[b] >= 0 && h_num_samples[b] >= 0 && h_num_samples[b] <= h_lengths[b],
               "cloud %lld: cannot draw %lld samples from %lld points", (long long)b, (long long)h_num_samples[b],
               (long long)h_lengths[b]);
    off[b + 1] = off[b] + (int32_t)h_lengths[b];
    soff[b + 1] = soff[b] + (int32_t)h_num_samples[b];
    st[b] = h_start_indices ? (int32_t)h_start_indices[b] : 0;
    nmax = std::max(nmax, h_lengths[b]);
  }
  GR_REQUIRE(off[batch] == n, "lengths do not sum to n");
  for (int64_t b = 0; b < batch; ++b)
    GR_REQUIRE(h_num_samples[b] == 0 || (st[b] >= 0 && (int64_t)st[b] < h_lengths[b]),
               "cloud %lld: start index %d outside [0, %lld)", (long long)b, st[b], (long long)h_lengths[b]);
  // G workgroups per cloud exchange candidates through memory, so all G*batch of them must be co-resident: one
  // 1024-thread workgroup per CU of THIS device (a CPX/DPX partition or a CU-masked stream has fewer than 256).
  int dev_id = 0, n_cu = 0;
  GR_HIP(hipGetDevice(&dev_id));
  GR_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id));
  GR_REQUIRE(n_cu > 0, "device reports no compute units");
  Carver c(ws);
  float* mind = c.take<float>(n);
  int32_t* d_off = c.take<int32_t>(batch + 1);
  int32_t* d_soff = c.take<int32_t>(batch + 1);
  int32_t* d_st = c.take<int32_t>(batch + 1);
  FpsCand* cand = c.take<FpsCand>((size_t)batch * 2 * FPS_GMAX);
  unsigned* arrive = c.take<unsigned>(batch + 1);  // [batch] = error flag
  unsigned long long* mslots = c.take<unsigned long long>((size_t)batch * 2 * FPS_GMAX * FPS_SLOT_STRIDE);
  uint64_t* mkeys_a = c.take<uint64_t>(n);
  uint64_t* mkeys_b = c.take<uint64_t>(n);
  int32_t* mvals = c.take<int32_t>(n);
  int32_t* mperm = c.take<int32_t>(n);
  float* spts = c.take<float>((size_t)n * 3);
  uint32_t* mbbox = c.take<uint32_t>(batch * 6);
  int32_t* mblk = c.take<int32_t>(batch + 1);
  const size_t sort_bytes = sort_pairs_temp_bytes(n);
  void* sort_tmp = c.take<char>(sort_bytes);
  const size_t ds_bytes = fps_ds_bytes(n, batch);
  void* ds_table = c.take<char>(ds_bytes);
  uint32_t* ds_range = c.take<uint32_t>(2 * batch);
  int32_t* ds_nvalid = c.take<int32_t>(batch + 1);
  int32_t* ds_ovf = c.take<int32_t>(batch + 1);  // [0]: a bucket did not fit
  bool curve_done = false;
  GR_HIP(hipMemcpyAsync(d_off, off.data(), sizeof(int32_t) * (batch + 1), hipMemcpyHostToDevice, stream));
  GR_HIP(hipMemcpyAsync(d_soff, soff.data(), sizeof(int32_t) * (batch + 1), hipMemcpyHostToDevice, stream));
  GR_HIP(hipMemcpyAsync(d_st, st.data(), sizeof(int32_t) * batch, hipMemcpyHostToDevice, stream));
  int* err = reinterpret_cast<int*>(arrive + batch);
  int G = (int)std::min<int64_t>({(int64_t)FPS_GMAX, (nmax + 2047) / 2048, std::max<int64_t>(1, (int64_t)n_cu / batch)});
  G = std::max(G, 1);
  // Attempt 0: G co-operating workgroups per cloud, launched co-operatively so that the runtime CHECKS the grid against
  // the device's residency instead of assuming it.  Attempt 1 (only if attempt 0 was refused or its inter-workgroup
  // exchange timed out, e.g. because another job holds CUs): one workgroup per cloud, which needs no co-residency.
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1) {
      if (G == 1) break;
      G = 1;
    }
    const int64_t per = ((nmax + G - 1) / G + FPS_T - 1) / FPS_T;  // points per thread
    GR_HIP(hipMemsetAsync(arrive, 0, sizeof(unsigned) * (batch + 1), stream));
    GR_HIP(hipMemsetAsync(cand, 0, sizeof(FpsCand) * (size_t)batch * 2 * FPS_GMAX, stream));
    GR_HIP(hipMemsetAsync(mslots, 0, sizeof(unsigned long long) * (size_t)batch * 2 * FPS_GMAX * FPS_SLOT_STRIDE, stream));
    bool launched = true;
    if (per <= 20 && !curve_done) {
      // curve order for the bucket pruning: boxes, codes, a sort over (cloud, code), gather
      std::vector<int32_t> h_blk(batch + 1);
      int rc = compute_bbox(points, off.data(), h_blk.data(), d_off, (int)batch, mbbox, mblk, stream);
      if (rc != GR_OK) return rc;
      const unsigned nblk = (unsigned)((n + 255) / 256);
      // The order only serves the pruning (results do not depend on it), and a wave's run is ~1 300 consecutive points: the top
      // 18 bits of the code (64 cells per axis) order the runs as well as all 30 do.
      constexpr int drop = 12, code_bits = 30 - drop;
      // Sort of (cloud, code, index).  Bucket sort (depth_sort.hip, the one grid_subsample uses: one pass over the top bits of
      // the code, every bucket finished in LDS -- 24 x 200 k points: 60 us) where the batch is not too ragged for its tables;
      // a cloud crowded into a few cells overflows a bucket and the call repeats the order with three radix passes (189 us).
      bool sorted = false;
      if (g_fps_bucket_sort.load() && nmax <= (1ll << 20) && batch * ((nmax + 2047) / 2048) <= fps_ds_rows(n, batch) &&
          depth_sort_table_bytes(nmax, (int)batch) <= ds_bytes) {
        std::vector<uint32_t> h_range(2 * batch);
        int max_dig = 3;
        for (int64_t b2 = 0; b2 < batch; ++b2) {
          // fewer top bits for small clouds (a bucket should hold a few hundred points), as long as the batch fills the device
          int dig = std::min(std::max(fps_bits_for((unsigned long long)h_lengths[b2]) - 9, 3), 9);
          while (dig < 9 && (batch << dig) < 8192) ++dig;
          h_range[2 * b2] = 1u;
          h_range[2 * b2 + 1] = (uint32_t)(code_bits - dig);
          max_dig = std::max(max_dig, dig);
        }
        GR_HIP(hipMemcpyAsync(ds_range, h_range.data(), sizeof(uint32_t) * 2 * batch, hipMemcpyHostToDevice, stream));
        GR_HIP(hipMemsetAsync(ds_ovf, 0, sizeof(int32_t), stream));
        uint32_t* field = reinterpret_cast<uint32_t*>(mkeys_a);  // [n]; the second half of the array: the sort's unused payload output
        hipLaunchKernelGGL(fps_curve_kernel, dim3(nblk), dim3(256), 0, stream, points, d_off, (int)batch, mbbox, (int)n, drop,
                           (unsigned long long*)nullptr, field);
        const DepthSortSegments sg{d_off, ds_range, nullptr, 0, nullptr, nullptr, nullptr, 1 << max_dig};
        rc = depth_sort_views(field, field, mkeys_b, mkeys_b, mvals, field + n, ds_nvalid, nmax, (int)batch, code_bits + 1, ds_table,
                              ds_bytes, stream, nullptr, 0, ds_ovf, 1, nullptr, &sg);
        if (rc != GR_OK) return rc;
        // (the gather is launched behind the sort at once; the overflow flag comes back with the synchronise that follows it)
        hipLaunchKernelGGL(fps_gather_kernel, dim3(nblk), dim3(256), 0, stream, points, d_off, (int)batch, mvals, (int)n, spts, mperm);
        int h_ovf = 0;
        GR_HIP(hipMemcpyAsync(&h_ovf, ds_ovf, sizeof(int), hipMemcpyDeviceToHost, stream));
        GR_HIP(hipStreamSynchronize(stream));  // (also: h_range and h_blk live on this stack frame)
        sorted = h_ovf == 0;
      }
      if (!sorted) {
        hipLaunchKernelGGL(fps_curve_kernel, dim3(nblk), dim3(256), 0, stream, points, d_off, (int)batch, mbbox, (int)n, drop,
                           reinterpret_cast<unsigned long long*>(mkeys_a), (uint32_t*)nullptr);
        int cloud_bits = 1;
        while ((1ll << cloud_bits) < batch) ++cloud_bits;
        rc = sort_pairs_u64_iota(sort_tmp, sort_bytes, mkeys_a, mkeys_b, (int64_t)1 << 40, mvals, n, 0, 32 + cloud_bits - drop, stream);
        if (rc != GR_OK) return rc;
        hipLaunchKernelGGL(fps_gather_kernel, dim3(nblk), dim3(256), 0, stream, points, d_off, (int)batch, mvals, (int)n, spts, mperm);
        GR_LAUNCH_CHECK();
        GR_HIP(hipStreamSynchronize(stream));  // h_blk lives on this stack frame
      }
      GR_LAUNCH_CHECK();
      curve_done = true;
    }
    {
      KernelTimer timer("fps", stream);
      const dim3 grid((unsigned)(batch * G)), block(FPS_T);
      const float* a_points = points;
      const float* a_spts = spts;
      const int32_t* a_perm = mperm;
      int a_G = G;
      void* multi_args[] = {&a_points, &a_spts, &a_perm, &d_off, &d_soff, &d_st, &mslots, &err, &a_G, &out_indices};
      void* single_args[] = {&a_points, &d_off, &d_soff, &d_st, &mind, &cand, &err, &a_G, &out_indices};
      const void* fn;
      void** args = multi_args;
      size_t lds = 0;  // the workgroup's original indices: PPT * 1024 ints
      // sixteen keys per workgroup when few workgroups share a cloud (large slabs): more candidates per round
      const bool m16 = G <= 16 && per > 10;
      if (per <= 4) fn = reinterpret_cast<const void*>(fps_multi_kernel<4, 8>), lds = 4;
      else if (per <= 7) fn = reinterpret_cast<const void*>(fps_multi_kernel<7, 8>), lds = 7;
      else if (per <= 10) fn = reinterpret_cast<const void*>(fps_multi_kernel<10, 8>), lds = 10;
      else if (per <= 13) fn = m16 ? reinterpret_cast<const void*>(fps_multi_kernel<13, 16>) : reinterpret_cast<const void*>(fps_multi_kernel<13, 8>), lds = 13;
      else if (per <= 16) fn = m16 ? reinterpret_cast<const void*>(fps_multi_kernel<16, 16>) : reinterpret_cast<const void*>(fps_multi_kernel<16, 8>), lds = 16;
      else if (per <= 20) fn = m16 ? reinterpret_cast<const void*>(fps_multi_kernel<20, 16>) : reinterpret_cast<const void*>(fps_multi_kernel<20, 8>), lds = 20;
      lds *= (size_t)FPS_T * sizeof(int);
      if (lds > 48 * 1024) GR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (per > 20) {  // slab too large for registers: one sample per round, distances streamed from L2
        fn = reinterpret_cast<const void*>(fps_kernel<0>);
        args = single_args;
      }
      hipError_t le;
      if (G > 1 && attempt == 0 && g_fps_force.load() == 1) le = hipErrorCooperativeLaunchTooLarge;  // test switch: "refused"
      else if (G > 1) le = hipLaunchCooperativeKernel(fn, grid, block, args, (unsigned)lds, stream);
      else le = hipLaunchKernel(fn, grid, block, args, lds, stream);
      if (le != hipSuccess) {
        (void)hipGetLastError();
        if (G > 1 && attempt == 0) launched = false;  // grid not co-resident on this device: fall back
        else {
          set_error("fps launch failed: %s", hipGetErrorString(le));
          return GR_ERR_HIP;
        }
      }
    }
    if (!launched) continue;
    int h_err = 0;
    GR_HIP(hipMemcpyAsync(&h_err, err, sizeof(int), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));  // also keeps the host staging vectors alive past the copies
    if (G > 1 && attempt == 0 && g_fps_force.load() == 2) h_err = 1;  // test switch: treat the co-operative run as timed out
    if (h_err == 0) return GR_OK;
    GR_REQUIRE(attempt == 0 && G > 1, "fps: exchange timed out with a single workgroup per cloud (internal error)");
  }
  set_error("fps: inter-workgroup exchange timed out and the single-workgroup retry was not possible");
  return GR_ERR_HIP;
}

extern "C" int gr_fps_debug_force_fallback(int mode) {
  const int old = gr::g_fps_force_get();
  if (mode >= 0 && mode <= 2) gr::g_fps_force_set(mode);
  return old;
}
