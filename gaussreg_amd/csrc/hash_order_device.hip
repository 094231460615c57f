// The iteration order of std::unordered_map<size_t, T> (libstdc++) after inserting distinct keys in a given order,
// computed ON THE DEVICE for many clouds at once -- the row order the reference emits subsampled points in
// (geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:28-47).  hash_order.hip replays the container's
// linking rules on one host thread per cloud (and stays as the checker / fallback); this file evaluates the same order in
// closed form, which is what makes it parallel:
//
//   libstdc++ keeps one singly linked list; a node inserted into an EMPTY bucket becomes the new list head, a node
//   inserted into a non-empty bucket goes right behind the bucket's before-node, i.e. to the FRONT of its bucket's group
//   (_M_insert_bucket_begin).  Hence at any time the list reads: groups in DEscending order of their creation time,
//   inside a group nodes in DEscending insertion time.  A rehash (_M_rehash_aux, unique keys) relinks every node, walking
//   the old list front to back, by exactly the same two rules -- so after a rehash the same statement holds with
//   "time" = position in the old list, and later insertions simply continue the clock.
//
// So with T(e) = the clock value of element e under the current table (old-list position for elements that lived through
// the last rehash, insertion index afterwards; T is a permutation of 0 .. m-1), bucket b(e) = key mod n, and
// first(b) = min T over the bucket, the list position of e is
//     pos(e) = #{ e' : first(b(e')) > first(b(e)) }  +  #{ e' in the same bucket : T(e') > T(e) }.
// The first term is a suffix sum over G[f] = (size of the group whose oldest element has clock f, else 0); the second is
// a walk over the (short) bucket chain.  One such evaluation per rehash of the container's history (the bucket counts
// come from the REAL _Prime_rehash_policy object of the libstdc++ this library is built against), plus one for the final
// table; every evaluation is a handful of data-parallel passes over all clouds.
#include <atomic>
#include <unordered_map>  // std::__detail::_Prime_rehash_policy
#include <vector>

#include "common.hpp"

namespace gr {
namespace {
std::atomic<int> g_ho_force_prescan{0};  // test hook: take the slab-table pre-scan of the > 4 M-clock stages at any size

struct HoCloud {    // per cloud, per stage
  int32_t begin;    // first element (global rank) of the cloud
  int32_t m;        // elements taking part in this stage; for a cloud whose history has ended (n == 0): its size
  uint32_t n;       // bucket count of the table in force; 0 = the cloud is done, its positions are only carried along
  int32_t toff;     // where the cloud's bucket table starts
  int32_t soff;     // where the cloud's slab totals start (one per HO_SLAB clock values; same array as the tables)
  int32_t ebase;    // the cloud's slice of this stage's bucket ids / chain links starts at ebase + begin
};

// what the last stage does with the final list positions: the permutation, and / or rows moved straight to their place
struct HoEmit {
  int32_t* perm;           // perm[begin + pos] = element (may be null)
  const float* rows_in;    // rows_out[begin + pos] = rows_in[row_of[element]] (3 floats; rows_out may be null)
  const int32_t* row_of;
  float* rows_out;
};
__device__ __forceinline__ void ho_emit(const HoEmit& em, int begin, int pos, int e) {
  if (em.perm) em.perm[begin + pos] = e;  // j-th iterated element of the cloud = the element at list position j
  if (em.rows_out) {
    const int64_t r = em.row_of[e], o = (int64_t)begin + pos;
    em.rows_out[3 * o] = em.rows_in[3 * r];
    em.rows_out[3 * o + 1] = em.rows_in[3 * r + 1];
    em.rows_out[3 * o + 2] = em.rows_in[3 * r + 2];
  }
}

constexpr int HO_T = 256;
constexpr int HO_SLAB = 1024;  // clock values per workgroup of the suffix sum

__device__ __forceinline__ int ho_cloud_of(const int32_t* __restrict__ begins, int nb, int e) {
  int lo = 0, hi = nb;  // begins has nb + 1 entries
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (begins[mid] <= e) lo = mid;
    else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(HO_T) void ho_init_kernel(int n, const int32_t* __restrict__ begins, int nb,
                                                       int32_t* __restrict__ Ta, int32_t* __restrict__ Tb,
                                                       int32_t* __restrict__ head, int nhead) {
  // every bucket of every stage's table starts empty (-1): cleared here rather than by a memset launch of its own
  for (int i = blockIdx.x * HO_T + threadIdx.x; i < nhead; i += gridDim.x * HO_T) head[i] = -1;
  const int e = blockIdx.x * HO_T + threadIdx.x;
  if (e >= n) return;
  Ta[e] = Tb[e] = e - begins[ho_cloud_of(begins, nb, e)];  // both ping-pong buffers: an element keeps T = insertion index until a stage covers it
}

// Stage kernels run on 2-D grids: blockIdx.y = cloud, blockIdx.x * HO_T + lane = local element; the grid's x extent is the
// largest m of the stage, so the early stages (13, 29, 59 ... elements per cloud) cost a launch each and nothing more.
// One device-scope atomic per element (returning global atomics are served behind the L2s and are what this step costs):
// the exchange threads the element onto its bucket's chain; the bucket's size and oldest clock, which used to be an
// atomicAdd and an atomicMin next to it, are read off the (short) chain by the two kernels that need them.
// ALL the big stages in one launch (blockIdx.z = stage): the buckets of a stage depend on the keys only, not on the clocks.
__global__ __launch_bounds__(HO_T) void ho_bucket_kernel(const HoCloud* __restrict__ st_all, int batch,
                                                         const uint64_t* __restrict__ keys, int32_t* __restrict__ bkt,
                                                         int32_t* __restrict__ head, int32_t* __restrict__ nxt) {
  const HoCloud s = st_all[blockIdx.z * batch + blockIdx.y];
  const int le = blockIdx.x * HO_T + threadIdx.x;
  if (le >= s.m || s.n == 0) return;
  const int e = s.begin + le;
  const int b = s.toff + (int)(keys[e] % (uint64_t)s.n);  // std::hash<size_t> is the identity, not cached
  bkt[s.ebase + e] = b;
  nxt[s.ebase + e] = atomicExch(&head[b], e);  // chain order is irrelevant: the walks below only count and take minima
}

// The same chains without a single global atomic, for tables of up to HO_SLABS_MAX x HO_BSLAB buckets: a workgroup owns a
// slab of HO_BSLAB consecutive buckets of one (stage, cloud) table with their heads in LDS, streams through ALL the keys of
// the cloud (coalesced, L2-resident: a cloud's keys are a few hundred KB) and threads the ones that hash into its slab --
// (slabs) x the key reads and modulo operations, in exchange for LDS exchanges instead of device-scope ones (64 x 75 k cells:
// 9.6 M returning global atomics were 450 us, most of the reference-order evaluation).
// (the bucket ids come from a launch of their own, ho_bucket_id_kernel: one modulo per element and stage, not one per slab)
constexpr int HO_BSLAB = 16384;   // 64 KB of heads
constexpr int HO_SLABS_MAX = 16;
constexpr int HO_BT = 1024;
__global__ __launch_bounds__(HO_T) void ho_bucket_id_kernel(const HoCloud* __restrict__ st_all, int batch,
                                                            const uint64_t* __restrict__ keys, int32_t* __restrict__ bkt) {
  const HoCloud s = st_all[blockIdx.z * batch + blockIdx.y];
  const int le = blockIdx.x * HO_T + threadIdx.x;
  if (le >= s.m || s.n == 0) return;
  const int e = s.begin + le;
  const uint64_t k = keys[e];
  // std::hash<size_t> is the identity; voxel keys fit 32 bits unless the grid wrapped
  const uint32_t b = (k >> 32) == 0ull ? (uint32_t)k % s.n : (uint32_t)(k % (uint64_t)s.n);
  bkt[s.ebase + e] = s.toff + (int)b;
}
__global__ __launch_bounds__(HO_BT) void ho_bucket_slab_kernel(const HoCloud* __restrict__ st_all, int batch,
                                                               const int32_t* __restrict__ bkt,
                                                               int32_t* __restrict__ head, int32_t* __restrict__ nxt) {
  __shared__ int32_t s_head[HO_BSLAB];
  const HoCloud s = st_all[blockIdx.z * batch + blockIdx.y];
  const uint32_t lo = blockIdx.x * (uint32_t)HO_BSLAB;
  if (s.n == 0 || s.m <= 0 || lo >= s.n) return;
  const uint32_t width = min((uint32_t)HO_BSLAB, s.n - lo);
  for (uint32_t i = threadIdx.x; i < width; i += HO_BT) s_head[i] = -1;
  __syncthreads();
  const int32_t* row = bkt + s.ebase + s.begin;
  int32_t* link = nxt + s.ebase + s.begin;
  const uint32_t first = (uint32_t)s.toff + lo;
#pragma unroll 4
  for (int le = threadIdx.x; le < s.m; le += HO_BT) {
    const uint32_t r = (uint32_t)row[le] - first;
    if (r < width) link[le] = atomicExch(&s_head[r], s.begin + le);  // (LDS) chain order is irrelevant, see ho_bucket_kernel
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < width; i += HO_BT) head[s.toff + lo + i] = s_head[i];
}

// Which (cloud, block of the cloud) a workgroup of a 1-D stage grid takes.  xcd != 0 (many clouds per call): workgroup b runs
// on XCD b % 8, and all workgroups of a cloud are dealt to ONE XCD -- the chain walks and the final row moves are random
// accesses inside the cloud's own few hundred KB (tables, clocks, rows), which then stay in that XCD's L2 instead of being
// spread over eight of them (a cloud's rows written from eight L2s leave as partial sectors).  Grid: 8 * ceil(batch / 8) *
// nbx workgroups.  xcd == 0: cloud-major, as a 2-D grid would be.
__device__ __forceinline__ bool ho_block(int nbx, int batch, int xcd, int& bx, int& cloud) {
  const int id = blockIdx.x;
  if (xcd) {
    const int slot = id >> 3, g = slot / nbx;
    cloud = g * 8 + (id & 7);
    bx = slot - g * nbx;
  } else {
    cloud = id / nbx;
    bx = id - cloud * nbx;
  }
  return cloud < batch;
}

__global__ __launch_bounds__(HO_T) void ho_group_kernel(const HoCloud* __restrict__ st, const int32_t* __restrict__ T,
                                                        const int32_t* __restrict__ bkt, const int32_t* __restrict__ head,
                                                        const int32_t* __restrict__ nxt, int32_t* __restrict__ G,
                                                        int2* __restrict__ FW, int nbx, int batch, int xcd) {
  int bx, cloud;
  if (!ho_block(nbx, batch, xcd, bx, cloud)) return;
  const HoCloud s = st[cloud];
  const int le = bx * HO_T + threadIdx.x;
  if (le >= s.m || s.n == 0) return;
  const int e = s.begin + le;
  const int t = T[e];
  int first = t, cnt = 0, within = 0;
  for (int p = head[bkt[s.ebase + e]]; p >= 0; p = nxt[s.ebase + p]) {
    const int tp = T[p];
    first = min(first, tp);
    within += tp > t ? 1 : 0;
    ++cnt;
  }
  G[s.begin + t] = (t == first) ? cnt : 0;  // T is a permutation of 0 .. m-1: every slot written once
  // what ho_rank_kernel needs of the chain -- its oldest clock and how many of its elements are younger than this one -- so
  // that the chain (three or four dependent random reads per element) is walked ONCE per stage, not twice
  FW[e] = make_int2(first, within);
}

// The exclusive suffix sum S[f] = sum of G over clock values > f, in two parts.  Here, one workgroup per slab of HO_SLAB
// clocks: the suffix sum INSIDE the slab (S) and the slab's total (slab[]); the totals of the slabs above are added by
// the reader (ho_rank_kernel builds that table of m / HO_SLAB entries in LDS).  One workgroup per CLOUD walking all
// slabs, as this was until round 4, took 69 us for the 150 k clocks of a 200 k-point cloud's last table -- a third of
// the whole reference-order call; adding the group sizes into the slab totals with atomics in ho_group_kernel (tried)
// took 219 us: ~600 device-scope atomics per address serialise.
__global__ __launch_bounds__(HO_SLAB) void ho_suffix_kernel(const HoCloud* __restrict__ st, const int32_t* __restrict__ G,
                                                            int32_t* __restrict__ slabs, int32_t* __restrict__ S) {
  __shared__ int s_w[HO_SLAB / WAVE];
  const HoCloud s = st[blockIdx.y];
  const int slab = blockIdx.x;
  if (s.n == 0 || slab * HO_SLAB >= s.m) return;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int f = slab * HO_SLAB + HO_SLAB - 1 - tid;  // thread 0 takes the largest clock of the slab
  const int g = f < s.m ? G[s.begin + f] : 0;
  const int incl = wave_incl_scan_add_dpp(g);
  if (lane == WAVE - 1) s_w[wv] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int i = 0; i < HO_SLAB / WAVE; ++i) {
    base += i < wv ? s_w[i] : 0;
    total += s_w[i];
  }
  if (f < s.m) S[s.begin + f] = base + incl - g;  // everything with a larger clock in this slab
  if (tid == 0) slabs[s.soff + slab] = total;
}

// Clouds with more slabs than ho_rank_kernel keeps in LDS (HO_TAB: 4 M clocks): one workgroup per cloud turns the slab
// totals into "total of the slabs above" in place, and the rank kernel reads that.
__global__ __launch_bounds__(1024) void ho_slabscan_kernel(const HoCloud* __restrict__ st, int32_t* __restrict__ slabs) {
  __shared__ int s_w[1024 / WAVE];
  __shared__ int s_carry;
  const HoCloud s = st[blockIdx.x];
  if (s.n == 0) return;
  const int nslab = (s.m + HO_SLAB - 1) / HO_SLAB;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int hi = nslab; hi > 0; hi -= 1024) {
    const int j = hi - 1 - (int)threadIdx.x;  // thread 0 takes the highest slab
    const int g = j >= 0 ? slabs[s.soff + j] : 0;
    const int incl = wave_incl_scan_add_dpp(g);
    if (lane == WAVE - 1) s_w[wv] = incl;
    __syncthreads();
    int base = s_carry;
    for (int i = 0; i < wv; ++i) base += s_w[i];
    if (j >= 0) slabs[s.soff + j] = base + incl - g;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = base + incl;
    __syncthreads();
  }
}

// reads the clocks of the whole bucket chain from T, writes the new positions to the OTHER ping-pong buffer
constexpr int HO_TAB = 4096;
// LAST: the positions are final -- emitted (permutation and / or rows) instead of written back.
template <bool PRESCANNED, bool LAST>
__global__ __launch_bounds__(HO_T) void ho_rank_kernel(const HoCloud* __restrict__ st, const int32_t* __restrict__ T,
                                                       const int2* __restrict__ FW,
                                                       const int32_t* __restrict__ S, const int32_t* __restrict__ slabs,
                                                       int32_t* __restrict__ T_out, HoEmit em, int nbx, int batch, int xcd) {
  __shared__ int s_above[PRESCANNED ? 1 : HO_TAB];
  __shared__ int s_w[HO_T / WAVE];
  int bx, cloud;
  if (!ho_block(nbx, batch, xcd, bx, cloud)) return;
  const HoCloud s = st[cloud];
  const int le = bx * HO_T + threadIdx.x;
  if (bx * HO_T >= s.m) return;  // workgroup-uniform
  const int e = s.begin + le;
  if (s.n == 0) {  // finished cloud: keep its final positions in the buffer the next stage reads
    if (le < s.m) {
      if (LAST) ho_emit(em, s.begin, T[e], e);
      else T_out[e] = T[e];
    }
    return;
  }
  if (!PRESCANNED) {  // s_above[j] = total of the slabs above slab j
    const int nslab = (s.m + HO_SLAB - 1) / HO_SLAB;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int carry = 0;
    for (int r0 = 0; r0 < nslab; r0 += HO_T) {
      const int j = nslab - 1 - r0 - (int)threadIdx.x;  // thread 0 takes the highest slab
      const int g = j >= 0 ? slabs[s.soff + j] : 0;
      const int incl = wave_incl_scan_add_dpp(g);
      if (lane == WAVE - 1) s_w[wv] = incl;
      __syncthreads();
      int base = carry;
      for (int i = 0; i < HO_T / WAVE; ++i) {
        base += i < wv ? s_w[i] : 0;
        carry += s_w[i];
      }
      if (j >= 0) s_above[j] = base + incl - g;
      __syncthreads();
    }
  }
  if (le >= s.m) return;
  const int2 fw = FW[e];  // (ho_group_kernel walked the chain)
  const int first = fw.x, within = fw.y;
  const int above = PRESCANNED ? slabs[s.soff + first / HO_SLAB] : s_above[first / HO_SLAB];
  const int pos = S[s.begin + first] + above + within;
  if (LAST) ho_emit(em, s.begin, pos, e);
  else T_out[e] = pos;
}

// The first stages of every history are tiny (tables of 13, 29, 59 ... 2357 buckets): one workgroup per cloud runs them
// back to back with everything in LDS -- one launch instead of four per stage.
constexpr int HO_SMALL = 2357;  // largest m (= bucket count) handled here
__global__ __launch_bounds__(1024) void ho_small_stages_kernel(const HoCloud* __restrict__ st_all, int batch, int nsmall,
                                                               const uint64_t* __restrict__ keys,
                                                               int32_t* __restrict__ T_out_a, int32_t* __restrict__ T_out_b,
                                                               int last, HoEmit em) {
  __shared__ int sT[HO_SMALL], sTn[HO_SMALL], sB[HO_SMALL], sNx[HO_SMALL], sG[HO_SMALL];
  __shared__ int sFirst[HO_SMALL], sCnt[HO_SMALL], sHead[HO_SMALL];
  __shared__ int s_w[1024 / WAVE];
  __shared__ int s_carry;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int begin = st_all[c].begin;
  int m_done = 0;
  // everything the stages read from memory, requested ONCE up front (a stage is a dozen barriers of LDS work; a dependent
  // trip to memory for its descriptor and another for its keys were most of its 3.4 us): the descriptors of all small
  // stages (<= HO_SMALL_STAGES) and the first HO_SMALL keys, three per thread in registers
  constexpr int HO_SMALL_STAGES = 16, KPT = (HO_SMALL + 1023) / 1024;
  __shared__ int s_m[HO_SMALL_STAGES], s_n[HO_SMALL_STAGES];
  if (tid < HO_SMALL_STAGES) {
    const bool have = tid < nsmall;
    const HoCloud sd = st_all[(have ? tid : 0) * batch + c];
    s_m[tid] = have ? sd.m : 0;
    s_n[tid] = have ? (int)sd.n : 0;
  }
  // (the last small stage's m -- or, for a cloud whose history ended earlier, its size -- bounds every index the stages use)
  const int m_all = nsmall > 0 ? st_all[(nsmall - 1) * batch + c].m : 0;
  uint64_t kreg[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) kreg[u] = m_all > 0 ? keys[begin + min(tid + u * 1024, m_all - 1)] : 0ull;
  for (int i = tid; i < HO_SMALL; i += 1024) sT[i] = i;
  __syncthreads();
  for (int k = 0; k < nsmall; ++k) {
    if (s_n[k] == 0) break;  // block-uniform: this cloud's history has ended
    const int m = s_m[k], n = s_n[k];
    for (int i = tid; i < n; i += 1024) {
      sFirst[i] = 0x7fffffff;
      sCnt[i] = 0;
      sHead[i] = -1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      const int e = tid + u * 1024;
      if (e >= m) break;
      const uint64_t kk = kreg[u];
      const int b = (kk >> 32) == 0ull ? (int)((uint32_t)kk % (uint32_t)n) : (int)(kk % (uint64_t)n);
      sB[e] = b;
      atomicMin(&sFirst[b], sT[e]);
      atomicAdd(&sCnt[b], 1);
      sNx[e] = atomicExch(&sHead[b], e);
    }
    __syncthreads();
    for (int e = tid; e < m; e += 1024) sG[sT[e]] = (sT[e] == sFirst[sB[e]]) ? sCnt[sB[e]] : 0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    // exclusive suffix sum of G over the clock, in place (G[f] <- sum over clocks > f)
    for (int hi = m; hi > 0; hi -= 1024) {
      const int f = hi - 1 - tid;
      const int g = f >= 0 ? sG[f] : 0;
      const int incl = wave_incl_scan_add_dpp(g);
      if (lane == WAVE - 1) s_w[wv] = incl;
      __syncthreads();
      int base = s_carry;
      for (int i = 0; i < wv; ++i) base += s_w[i];
      if (f >= 0) sG[f] = base + incl - g;
      __syncthreads();
      if (tid == 1023) s_carry = base + incl;
      __syncthreads();
    }
    for (int e = tid; e < m; e += 1024) {
      const int b = sB[e], t = sT[e];
      int within = 0;
      for (int p = sHead[b]; p >= 0; p = sNx[p]) within += sT[p] > t ? 1 : 0;
      sTn[e] = sG[sFirst[b]] + within;
    }
    __syncthreads();
    for (int e = tid; e < m; e += 1024) sT[e] = sTn[e];
    m_done = m;
    __syncthreads();
  }
  if (last) {  // no cloud has a later stage: m_done is the whole cloud and the positions are final
    for (int e = tid; e < m_done; e += 1024) ho_emit(em, begin, sT[e], begin + e);
    return;
  }
  for (int e = tid; e < m_done; e += 1024) T_out_a[begin + e] = T_out_b[begin + e] = sT[e];
}

}  // namespace

// (element count at which the container rehashes, new bucket count) for a cloud of m distinct keys
static void rehash_schedule(int64_t m, std::vector<std::pair<int64_t, uint64_t>>& out) {
  out.clear();
  std::__detail::_Prime_rehash_policy policy;
  std::size_t bkt = 1;
  int64_t i = 0;
  while (i < m) {
    const auto need = policy._M_need_rehash(bkt, static_cast<std::size_t>(i), 1);
    if (need.first) {
      out.emplace_back(i, need.second);
      bkt = need.second;
    }
    // _M_need_rehash answers "no" without touching its state until n_elt + 1 exceeds _M_next_resize
    const int64_t nxt = static_cast<int64_t>(policy._M_next_resize);
    i = nxt > i ? nxt : i + 1;
  }
}

size_t hash_order_device_bytes(int64_t n, int64_t batch) {
  // worst case bucket table: the policy at most doubles past the element count (+ the prime gap): 4 n + slack per cloud
  const size_t buckets = (size_t)(4 * n + 64 * batch + 64) + (size_t)(n / 256 + 64 * batch + 64);  // tables + slab sums
  // bucket ids and chain links per (stage, cloud): the stages' element counts at least halve going back from the last
  // two (m, < m, < m / 2, ...): < 3 n in all
  const size_t links = (size_t)(3 * n + 64 * batch + 64);
  return align_up((size_t)n * 4, 256) * 4 + align_up((size_t)n * 8, 256) + align_up(links * 4, 256) * 2 + align_up(2 * buckets * 4, 256) + align_up((size_t)(batch + 1) * 4, 256) * 2 +
         align_up((size_t)batch * sizeof(HoCloud), 256) * 64 + 4096;
}

// keys: device, n distinct-per-cloud keys in insertion order, clouds contiguous (h_begins: batch + 1 host offsets).
// perm_out (device, n): perm_out[begin_c + j] = global index of the j-th element the container would iterate.
// rows_out (optional, with rows_in / row_of): rows_out[begin_c + j] = rows_in[row_of[that element]] (3 floats per row) --
// the caller's gather by the permutation, done by the launch that knows the final positions.  perm_out may then be null.
int hash_order_device(const uint64_t* keys, const int64_t* h_begins, int64_t batch, int32_t* perm_out, void* ws,
                      size_t ws_bytes, hipStream_t stream, const float* rows_in, const int32_t* row_of, float* rows_out) {
  const int64_t n = h_begins[batch];
  if (n == 0) return GR_OK;
  GR_REQUIRE(n < (1ll << 30) && batch >= 1, "hash_order_device: bad sizes");
  // ---- schedules
  std::vector<std::vector<std::pair<int64_t, uint64_t>>> sched(batch);
  std::vector<int32_t> begins(batch + 1);
  size_t nstage = 0;
  for (int64_t c = 0; c < batch; ++c) {
    begins[c] = (int32_t)h_begins[c];
    rehash_schedule(h_begins[c + 1] - h_begins[c], sched[c]);
    nstage = std::max(nstage, sched[c].size());
    const uint64_t nfinal = sched[c].empty() ? 1 : sched[c].back().second;
    GR_REQUIRE(nfinal < (1ull << 30), "hash_order_device: bucket count out of range");
  }
  begins[batch] = (int32_t)n;
  GR_REQUIRE(nstage <= 60, "hash_order_device: too many rehash stages");
  // stage k (k = 0 .. nstage-1): the table installed by rehash k, evaluated over the elements present when rehash k+1
  // strikes (or all of them for a cloud's last table)
  // every (stage, cloud) gets its own slice of the bucket tables (the bucket counts roughly double from stage to stage, so
  // all slices together are about twice the final tables): ONE clear up front instead of one per stage
  std::vector<HoCloud> hs(nstage * batch);
  int64_t table_entries = 0, link_entries = 0;
  for (size_t k = 0; k < nstage; ++k)
    for (int64_t c = 0; c < batch; ++c) {
      HoCloud s{begins[c], (int32_t)(h_begins[c + 1] - h_begins[c]), 0u, 0, 0, 0};
      const auto& sc = sched[c];
      if (k < sc.size()) {
        s.n = (uint32_t)sc[k].second;
        s.m = (int32_t)(k + 1 < sc.size() ? sc[k + 1].first : h_begins[c + 1] - h_begins[c]);
        s.toff = (int32_t)table_entries;
        table_entries += s.n;
        s.soff = (int32_t)table_entries;
        table_entries += (s.m + HO_SLAB - 1) / HO_SLAB;
      }
      hs[k * batch + c] = s;
    }
  // stages whose largest m fits the LDS kernel (the rehash thresholds are the same prime sequence for every cloud)
  size_t nsmall = 0;
  while (nsmall < nstage) {
    int64_t mm = 0, nn = 0;
    for (int64_t c = 0; c < batch; ++c)
      if (hs[nsmall * batch + c].n) {
        mm = std::max<int64_t>(mm, hs[nsmall * batch + c].m);
        nn = std::max<int64_t>(nn, hs[nsmall * batch + c].n);
      }
    if (mm > HO_SMALL || nn > HO_SMALL) break;
    ++nsmall;
  }
  // the stages after those run one launch per step over all clouds; each (stage, cloud) has its own slice of bucket ids and
  // chain links so that ONE launch can thread the chains of all of them
  for (size_t k = nsmall; k < nstage; ++k)
    for (int64_t c = 0; c < batch; ++c) {
      HoCloud& s = hs[k * batch + c];
      if (s.n == 0) continue;
      s.ebase = (int32_t)(link_entries - begins[c]);
      link_entries += s.m;
    }
  GR_REQUIRE(table_entries < (1ll << 31) && link_entries < (1ll << 31), "hash_order_device: bucket tables out of range");
  const size_t buckets = (size_t)table_entries;
  Carver cv(ws);
  int32_t* Ta = cv.take<int32_t>(n);
  int32_t* Tb = cv.take<int32_t>(n);
  int32_t* bkt = cv.take<int32_t>(link_entries);
  int32_t* nxt = cv.take<int32_t>(link_entries);
  int32_t* G = cv.take<int32_t>(n);
  int2* FW = cv.take<int2>(n);
  int32_t* S = cv.take<int32_t>(n);
  int32_t* head = cv.take<int32_t>(buckets);
  int32_t* d_begins = cv.take<int32_t>(batch + 1);
  HoCloud* d_st = cv.take<HoCloud>(nstage * batch);
  GR_REQUIRE(ws && cv.used() <= ws_bytes, "hash_order_device: workspace too small (%zu > %zu)", cv.used(), ws_bytes);
  // The two small tables go up through pinned per-thread staging; an event says when the copies have left it, so the
  // call returns without a stream synchronise and the next call on this thread waits (normally not at all) before reuse.
  static thread_local hipEvent_t staged = nullptr;
  if (staged == nullptr) GR_HIP(hipEventCreateWithFlags(&staged, hipEventDisableTiming));
  else GR_HIP(hipEventSynchronize(staged));
  const size_t begins_bytes = sizeof(int32_t) * (batch + 1), hs_bytes = sizeof(HoCloud) * hs.size();
  char* stage = static_cast<char*>(pinned_scratch(4, align_up(begins_bytes, 256) + hs_bytes));
  GR_REQUIRE(stage != nullptr, "hash_order_device: pinned staging buffer could not be allocated");
  memcpy(stage, begins.data(), begins_bytes);
  memcpy(stage + align_up(begins_bytes, 256), hs.data(), hs_bytes);
  // d_begins and d_st are neighbours in the workspace with the staging buffer's layout: one copy
  GR_REQUIRE(reinterpret_cast<char*>(d_st) == reinterpret_cast<char*>(d_begins) + align_up(begins_bytes, 256),
             "hash_order_device: workspace layout");
  GR_HIP(hipMemcpyAsync(d_begins, stage, align_up(begins_bytes, 256) + hs_bytes, hipMemcpyHostToDevice, stream));
  GR_HIP(hipEventRecord(staged, stream));
  const dim3 blk(HO_T), grd((unsigned)((n + HO_T - 1) / HO_T));
  hipLaunchKernelGGL(ho_init_kernel, grd, blk, 0, stream, (int)n, d_begins, (int)batch, Ta, Tb, head, (int)buckets);
  int32_t* Tin = Ta;
  int32_t* Tout = Tb;
  const HoEmit em{perm_out, rows_in, row_of, rows_out};
  // stage 0 (13 buckets) is always small: with no big stage the LDS kernel emits
  GR_REQUIRE(nsmall <= 16, "hash_order_device: more small stages than the LDS kernel keeps descriptors for");
  hipLaunchKernelGGL(ho_small_stages_kernel, dim3((unsigned)batch), dim3(1024), 0, stream, d_st, (int)batch, (int)nsmall, keys,
                     Ta, Tb, nsmall == nstage ? 1 : 0, em);
  const bool prescan_always = g_ho_force_prescan.load() != 0;  // test hook (gr_hash_order_debug_force_prescan): the > 4 M-clock path
  int64_t big_m = 0;
  for (size_t k = nsmall; k < nstage; ++k)
    for (int64_t c = 0; c < batch; ++c)
      if (hs[k * batch + c].n) big_m = std::max<int64_t>(big_m, hs[k * batch + c].m);
  if (nsmall < nstage && big_m > 0) {
    uint64_t big_n = 0;
    for (size_t k = nsmall; k < nstage; ++k)
      for (int64_t c = 0; c < batch; ++c) big_n = std::max<uint64_t>(big_n, hs[k * batch + c].n);
    const unsigned slabs = (unsigned)((big_n + HO_BSLAB - 1) / HO_BSLAB);
    // (a few clouds: a few hundred thousand atomics are over before the slab workgroups have streamed through their clouds --
    // one 200 k cloud measured 0.20 ms with the atomics, 0.25 ms with the slabs)
    if (slabs <= (unsigned)HO_SLABS_MAX && link_entries >= (1 << 20)) {
      hipLaunchKernelGGL(ho_bucket_id_kernel, dim3((unsigned)((big_m + HO_T - 1) / HO_T), (unsigned)batch, (unsigned)(nstage - nsmall)),
                         blk, 0, stream, d_st + nsmall * batch, (int)batch, keys, bkt);
      hipLaunchKernelGGL(ho_bucket_slab_kernel, dim3(slabs, (unsigned)batch, (unsigned)(nstage - nsmall)), dim3(HO_BT), 0, stream,
                         d_st + nsmall * batch, (int)batch, bkt, head, nxt);
    } else  // (very large clouds: every slab would stream through all the bucket ids)
      hipLaunchKernelGGL(ho_bucket_kernel, dim3((unsigned)((big_m + HO_T - 1) / HO_T), (unsigned)batch, (unsigned)(nstage - nsmall)),
                         blk, 0, stream, d_st + nsmall * batch, (int)batch, keys, bkt, head, nxt);
  }
  for (size_t k = nsmall; k < nstage; ++k) {
    const HoCloud* st = d_st + k * batch;
    int64_t max_m = 0;  // every cloud takes part: finished ones have their positions carried along (or emitted)
    for (int64_t c = 0; c < batch; ++c) max_m = std::max<int64_t>(max_m, hs[k * batch + c].m);
    const bool last = k + 1 == nstage;
    if (max_m == 0) continue;
    const int nbx = (int)((max_m + HO_T - 1) / HO_T);
    const int xcd = batch >= 32 ? 1 : 0;  // (few clouds: a cloud per XCD would leave XCDs idle)
    const dim3 eg((unsigned)(nbx * (xcd ? (batch + 7) / 8 * 8 : batch)));
    hipLaunchKernelGGL(ho_group_kernel, eg, blk, 0, stream, st, Tin, bkt, head, nxt, G, FW, nbx, (int)batch, xcd);
    const int64_t nslab = (max_m + HO_SLAB - 1) / HO_SLAB;
    hipLaunchKernelGGL(ho_suffix_kernel, dim3((unsigned)nslab, (unsigned)batch), dim3(HO_SLAB), 0, stream, st, G, head, S);
    const bool pre = nslab > HO_TAB || prescan_always;
    if (pre) hipLaunchKernelGGL(ho_slabscan_kernel, dim3((unsigned)batch), dim3(1024), 0, stream, st, head);
#define GR_HO_RANK(P, L) \
  hipLaunchKernelGGL((ho_rank_kernel<P, L>), eg, blk, 0, stream, st, Tin, FW, S, head, Tout, em, nbx, (int)batch, xcd)
    if (pre) {
      if (last) GR_HO_RANK(true, true); else GR_HO_RANK(true, false);
    } else {
      if (last) GR_HO_RANK(false, true); else GR_HO_RANK(false, false);
    }
#undef GR_HO_RANK
    std::swap(Tin, Tout);
  }
  GR_LAUNCH_CHECK();
  return GR_OK;
}

}  // namespace gr

// Test hook: the device evaluation on its own (keys on the device, offsets on the host).
extern "C" size_t gr_hash_order_device_workspace_bytes(int64_t n, int64_t batch) {
  if (n < 0 || batch < 0) return 0;
  return gr::hash_order_device_bytes(n, batch);
}

extern "C" int gr_hash_order_device(const uint64_t* d_keys, const int64_t* h_begins, int64_t batch, int32_t* d_perm,
                                    void* ws, size_t ws_bytes, void* stream) {
  GR_REQUIRE(batch >= 0 && (batch == 0 || h_begins), "bad arguments");
  if (batch == 0) return GR_OK;
  GR_REQUIRE(h_begins[0] == 0, "h_begins must start at 0");
  for (int64_t c = 0; c < batch; ++c) GR_REQUIRE(h_begins[c + 1] >= h_begins[c], "h_begins must be non-decreasing");
  GR_REQUIRE(h_begins[batch] == 0 || (d_keys && d_perm), "null argument");
  return gr::hash_order_device(d_keys, h_begins, batch, d_perm, ws, ws_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr,
                               nullptr);
}

extern "C" int gr_hash_order_debug_force_prescan(int on) {
  const int old = gr::g_ho_force_prescan.load();
  if (on == 0 || on == 1) gr::g_ho_force_prescan.store(on);
  return old;
}
