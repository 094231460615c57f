// Thread-per-query radius search, candidates straight through the vector L1 (round 6).  Included by radius_neighbors.hip
// inside namespace gr::{anonymous} (it uses that file's BatchGrid / cell_coord / band conventions).
//
// Same semantics as every other search kernel of this library (radius_neighbors_cpu.cpp:3-91 with nanoflann's metric,
// strict `<` and ascending order: nanoflann.hpp:423-446, 249-253, 1287), different mapping:
//   * a workgroup is ONE wave of 64 consecutive cell-ordered queries and a THREAD owns a query from its first distance test
//     to its finished row -- no decode / keys re-mapping / ranking phases (60 % of the three-threads-per-query kernels);
//   * the candidates are NOT staged in LDS.  Lanes are neighbouring queries, so at any step the 64 lanes read a handful of
//     lines of the cell-sorted coordinate planes (x[], y[], z[] written by fine_kernel next to its float4 records); each
//     thread pulls four consecutive candidates per plane with one 16-byte load (dword aligned) and tests them two at a time
//     with packed fp32 math.  Round 5's version of this mapping staged 12 KB of planes per wave and was left with 6 waves per
//     CU; here LDS holds only the hit lists (u16 codes: band << 12 | offset from the wave's first candidate of that band),
//     the threads' range tables and band bases and a small key scratch: 10.5 KB per wave;
//   * sorting: the first NET hits -> one 32-bit word per hit, (fixed-point distance << log2 NET) | list slot, sorted by
//     Batcher's odd-even merge network with v_min_u32 / v_max_u32 (2 instructions per comparator; the 64-bit (distance,
//     index) words of round 5 cost 6).  The fixed-point distance is a monotone map of the fp32 distance (d * 2^FB / r^2,
//     truncated), so words of different distance fields are in the reference's order; two hits whose fields are EQUAL (exact
//     ties, or distances closer than r^2 2^-27) are detected after the sort and that query goes the exact way below;
//   * a query with more than NET hits, or with an equal pair of distance fields, is finished by the whole wave: lanes =
//     its candidates, hits compacted by ballot into (distance bits << 32 | index) words, ranked by counting;
//   * rows: transposed through LDS in blocks of 16 columns and written as 128-byte pieces.  DIRECT: int64 rows of the
//     caller's width in the caller's order; otherwise compact u32 rows, scattered into the caller's order as whole sectors,
//     which tq_expand_kernel widens to int64 rows once the host knows the width (the bare radius_neighbors, whose width is the
//     largest count);
//   * PRESEL (rows of a known width far below the hit counts): a histogram pre-selection in front of the network, see the
//     comment at the tests.
// A workgroup that cannot finish raises its flag (2: a single range beyond 12 bits, 3: more than 40 of its 64 queries beyond
// the network, 5: a query with more hits than the key scratch) and the caller repeats the call on count + fill.  The number of wave-finished queries is reported: a call in
// which they are more than an eighth of all queries is complete, but its call site starts on the next kernel next time.
#pragma once
#include <type_traits>

constexpr int TQ_ROW_CAP = 64;    // the widest row the expand kernel can deliver: two halves of TQ_ROW_HALF u32 slots per query
constexpr int TQ_ROW_HALF = 32;   // (first halves dense in one array -- 128 bytes per query; second halves, only written for
                                  // wave-finished queries, in another)
constexpr int TQ_BKEYS = 192;     // hits of a wave-finished query
constexpr int TQ_HOPELESS = 40;   // queries beyond the network in one wave at which the workgroup gives up instead
constexpr int TQ_EXT_MAX = 4095;  // band extent of a wave (offsets are 12 bits)

template <int N>
struct TqNetwork {  // Batcher's odd-even merge sort for N = 2^k inputs (Knuth 5.2.2 M)
  static constexpr int CAP = N == 64 ? 543 : (N == 32 ? 191 : (N == 16 ? 63 : 19));
  unsigned char a[CAP], b[CAP];
  int n;
  constexpr TqNetwork() : a{}, b{}, n(0) {
    for (int p = 1; p < N; p *= 2)
      for (int k = p; k >= 1; k /= 2)
        for (int j = k % p; j <= N - 1 - k; j += 2 * k)
          for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i)
            if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
              a[n] = (unsigned char)(i + j);
              b[n] = (unsigned char)(i + j + k);
              ++n;
            }
  }
};

typedef float tq_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef long long tq_ll2 __attribute__((ext_vector_type(2), aligned(8)));
// a - b on two lanes (one v_pk_add_f32; written as `a - b` the compiler splits half of the loop's subtractions in two)
__device__ __forceinline__ f32x2 tq_pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int N>
static constexpr TqNetwork<N> tq_network{};

template <int NET, bool DIRECT>
struct TqLds {
  static constexpr int LIST_ROWS = NET + 5;  // the cursor is clamped to NET + 1 once per step of four candidates
  static constexpr int TABLE_MAX = 32;       // clouds whose offsets / grids are staged for the set-up (in the lists' place)
  static constexpr int RB = 16;              // columns of the row buffer: the rows leave in blocks of 16 columns (128 bytes)
  static constexpr int RS = RB + 1;          // its row stride (odd: conflict-free column writes)
  struct Search {
    unsigned short lists[LIST_ROWS * WAVE];
    unsigned short trng[2 * (NBAND + 1) * WAVE];  // the threads' non-empty ranges (start code, end code) + an empty one
    // (NET = 64 keeps no idx32: 16 KB per wave would leave 8 waves per CU; its indices are gathered again after the sort)
    char idx_room[(NET <= 32 && NET * WAVE * 4 > (LIST_ROWS + 2 * (NBAND + 1)) * WAVE * 2) ? NET * WAVE * 4 - (LIST_ROWS + 2 * (NBAND + 1)) * WAVE * 2 : 1];
  };  // (after the tests the same bytes hold idx32[NET][64]: the support index of every list slot)
  struct Rows {  // once the sorted indices are in registers the lists are dead: rows are transposed here
    unsigned int rowbuf[WAVE * RS];
    int2 qinfo[WAVE];  // (original index, count or -1 = finished by the wave)
  };
  union {
    Search s;
    Rows r;
    unsigned long long bkeys[TQ_BKEYS];  // wave-finished queries (after the rows have left)
    char tables[(TABLE_MAX + 1) * 4 + 16 + TABLE_MAX * 64];
  };
  int lbase[NBAND * WAVE];  // per thread and band: first candidate - (band << 12)
};
static_assert(sizeof(TqLds<32, true>) <= 11 * 1024 && sizeof(TqLds<64, true>) <= 19 * 1024, "tq kernel: LDS per wave");

template <int NET, bool DIRECT, bool PRESEL = false>
__global__ __launch_bounds__(WAVE) void tq_kernel(
    const float4* __restrict__ sorted_q, int nq, const int32_t* __restrict__ q_off, int nb,
    const BatchGrid* __restrict__ grids, const int32_t* __restrict__ start_s, const float4* __restrict__ sorted_s,
    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz, int ns_total, float r2,
    int32_t* __restrict__ blk_stats, int width, int64_t pad_value, int64_t* __restrict__ out, uint32_t* __restrict__ rows32,
    int32_t* __restrict__ q_cnt, size_t rows_hi, int mono, int stop) {
#define TQ_STOP(K, V) if (stop == (K)) { if ((V) == 0x7fffffff) blk_stats[0] = 1; return; }
  using L = TqLds<NET, DIRECT>;
  constexpr int SB = NET == 64 ? 6 : (NET == 32 ? 5 : 4);        // slot bits of a sort word
  constexpr int FB = 32 - SB;                                     // distance field
  constexpr unsigned FIX_MAX = (1u << FB) - 1u - (unsigned)NET;   // real hits stay below the pad words
  static_assert(!PRESEL || (DIRECT && NET == 64), "the pre-selection is built for rows of a known width on the 64-hit network");
  __shared__ __attribute__((aligned(16))) L lds;
  unsigned short* lists = lds.s.lists;
  unsigned short* trng = lds.s.trng;
  int* lbase = lds.lbase;
  const int tcap = nb <= L::TABLE_MAX ? nb : 0;
  int* s_qoff = reinterpret_cast<int*>(lds.tables);
  BatchGrid* s_grids = reinterpret_cast<BatchGrid*>(lds.tables + ((size_t)(tcap + 1) * 4 + 15) / 16 * 16);

  const int lane = threadIdx.x;
  const int nblk = (nq + WAVE - 1) / WAVE;
  const int per_xcd = gridDim.x / 8;
  const int blk = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // one contiguous eighth of the cell-ordered queries per XCD
  if (blk >= nblk) return;
  const int t = blk * WAVE + lane;
  const bool valid = t < nq;
  if (tcap > 0) {
    for (int i = lane; i <= nb; i += WAVE) s_qoff[i] = q_off[i];
    const int4* gsrc = reinterpret_cast<const int4*>(grids);
    int4* gdst = reinterpret_cast<int4*>(s_grids);
    for (int i = lane; i < nb * 4; i += WAVE) gdst[i] = gsrc[i];
  }
  float4 qp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) qp = sorted_q[t];
  const int orig = valid ? __float_as_int(qp.w) : 0;
  __syncthreads();
  // ---- set-up: the nine candidate ranges (positions in the cell-sorted supports), band k = 3 * (dz + 1) + (dy + 1)
  int p0[NBAND], p1[NBAND];
#pragma unroll
  for (int k = 0; k < NBAND; ++k) p0[k] = p1[k] = 0;
  if (valid) {
    int b;
    BatchGrid g;
    if (tcap > 0) {
      b = find_batch(s_qoff, nb, orig);
      g = s_grids[b];
    } else {
      b = find_batch(q_off, nb, orig);
      g = grids[b];
    }
    const double ux = cell_coord(qp.x, g.org[0], g.inv_cell_x), kx = (double)g.xk;
    const double uy = cell_coord(qp.y, g.org[1], g.inv_cell);
    const double uz = cell_coord(qp.z, g.org[2], g.inv_cell);
    const double tx = (double)(g.dim[0] - 1), ty = (double)(g.dim[1] - 1), tz = (double)(g.dim[2] - 1);
    if ((ux + kx >= 0.0) && (ux - kx <= tx)) {  // NaN coordinates: no candidates
      const int lx = (int)fmin(fmax(ux - kx, 0.0), tx);
      const int hx = (int)fmin(fmax(ux + kx, 0.0), tx);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double cz = uz + (double)(j - 1);
        if (cz >= 0.0 && cz <= tz) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double cy = uy + (double)(i - 1);
            if (cy >= 0.0 && cy <= ty) {
              const int base = g.cell_base + g.dim[0] * ((int)cy + g.dim[1] * (int)cz);
              p0[3 * j + i] = start_s[base + lx];
              p1[3 * j + i] = start_s[base + hx + 1];
            }
          }
        }
      }
    }
  }
  TQ_STOP(1, p0[0] + p1[8] + p0[4])
  __syncthreads();  // the per-cloud tables (in the lists' place) are dead
  // ---- the thread's non-empty ranges as 16-bit codes (band << 12 | offset from the thread's OWN first candidate of the
  //      band); lbase[k][lane] + code = position in the planes.  (Offsets from the WAVE's first candidate of a band -- a
  //      uniform base table -- overflowed 12 bits on real scans: 64 consecutive queries of a sparse slab (walls) can span
  //      thirty rows of cells, and the band one slab below (the floor) then covers thirty full rows of a dense one.)
  int blk_flag = 0;
  int nrng = 0;
#pragma unroll
  for (int k = 0; k < NBAND; ++k) {
    const bool has = p1[k] > p0[k];
    if (__any(has && p1[k] - p0[k] > TQ_EXT_MAX)) blk_flag = 2;  // a single range beyond 12 bits (thousands of points per cell)
    lbase[k * WAVE + lane] = p0[k] - (k << 12);
    if (has) {
      const int code = k << 12;
      trng[(2 * nrng) * WAVE + lane] = (unsigned short)code;
      trng[(2 * nrng + 1) * WAVE + lane] = (unsigned short)(code + (p1[k] - p0[k]));
      ++nrng;
    }
  }
  trng[(2 * nrng) * WAVE + lane] = 0;  // the empty range a finished thread stays in
  trng[(2 * nrng + 1) * WAVE + lane] = 0;
  // (a wave's LDS operations are served in order: the reads below see these writes)
  const unsigned r2b = r2 == r2 ? __float_as_uint(r2) : 0u;  // NaN radius: nothing is a neighbour
  int n = 0;
  const float scale = (float)(1u << FB) / r2;  // distance -> fixed point (the sort words; PRESEL: the histogram bins)
  // PRESEL (rows truncated to `width` <= 56 where most queries have MORE than NET hits: the coarsest pyramid levels, ~150 hits
  // of ~770 candidates): the tests run twice.  Pass 1 counts the hits and drops each into a 64-bin histogram of its
  // fixed-point distance (one fire-and-forget ds_add_u32 per candidate on the thread's own column, two 16-bit bins per
  // word, in the lists' place); the thread then walks its bins to the first one at which the running count reaches `width`
  // and pass 2 lists only the hits up to and including that bin -- at least `width`, nearly always at most NET of them, and
  // every hit left out is farther than every hit kept (the bins are a monotone function of the fp32 distance, equal
  // distances share a bin).  From there on the query is an ordinary one; a crowded last bin (more than NET kept) or an equal
  // pair sends it to the wave's exact path, restricted to the same bins.
  int ntrue = 0;          // PRESEL: all hits of the query (what the reference would count)
  float fcut = INFINITY;  // PRESEL: a hit is listed iff d * scale < fcut
  if (!blk_flag) {
  auto scan = [&](auto pass_tag) {
    constexpr int PASS = decltype(pass_tag)::value;  // 0: list every hit; 1: count + histogram; 2: list the hits below fcut
    unsigned int* hist = reinterpret_cast<unsigned int*>(lists);
    // ---- tests: one flattened loop over the thread's ranges, four candidates per step, the planes of step i + 1 requested
    //      before step i is evaluated.  A hit appends its code to the thread's list (slot-major: entry i of lane l at
    //      [i][l]); the code is written to slot n UNCONDITIONALLY and n moves on only for a hit -- no branch per candidate
    const f32x2 qx = {qp.x, qp.x}, qy = {qp.y, qp.y}, qz = {qp.z, qp.z};
    unsigned short* my = lists + lane;
    int code = trng[lane], ecode = trng[WAVE + lane], badj = lbase[(code >> 12) * WAVE + lane];
    int kk = min(1, nrng);
    int ncode = trng[(2 * kk) * WAVE + lane], necode = trng[(2 * kk + 1) * WAVE + lane], nbadj = lbase[(ncode >> 12) * WAVE + lane];
    kk = min(2, nrng);
    // (32-bit byte offsets from uniform plane bases: the loads take the "saddr + voffset" form, no 64-bit address arithmetic)
    const char* const bx_ = reinterpret_cast<const char*>(px);
    const char* const by_ = reinterpret_cast<const char*>(py);
    const char* const bz_ = reinterpret_cast<const char*>(pz);
#define TQ_LOAD(X, Y, Z, OFF)                                   \
  X = *reinterpret_cast<const tq_f4u*>(bx_ + (OFF));            \
  Y = *reinterpret_cast<const tq_f4u*>(by_ + (OFF));            \
  Z = *reinterpret_cast<const tq_f4u*>(bz_ + (OFF));
    // one step: request the planes of the NEXT step into (NX, NY, NZ), then evaluate (X, Y, Z).  Moving on to the next range
    // is branch-free: every lane reads table row kk, the lanes that switch keep it (a divergent branch here ran in every
    // step anyway -- 64 lanes x 9 switches over ~40 steps -- and cost a dozen register moves)
#define TQ_STEP(X, Y, Z, NX, NY, NZ)                                                                                     \
  {                                                                                                                      \
    const int c4 = code + 4;                                                                                             \
    const bool sw = c4 >= ecode;                                                                                         \
    const int c2 = sw ? ncode : c4, e2 = sw ? necode : ecode, b2 = sw ? nbadj : badj;                                    \
    {                                                                                                                    \
      const int rc = trng[(2 * kk) * WAVE + lane], re = trng[(2 * kk + 1) * WAVE + lane];                                \
      const int rb = lbase[(rc >> 12) * WAVE + lane];                                                                    \
      ncode = sw ? rc : ncode;                                                                                           \
      necode = sw ? re : necode;                                                                                         \
      nbadj = sw ? rb : nbadj;                                                                                           \
      kk = sw ? min(kk + 1, nrng) : kk;                                                                                  \
    }                                                                                                                    \
    const unsigned off2 = c2 < e2 ? (unsigned)(b2 + c2) << 2 : 0u;                                                       \
    TQ_LOAD(NX, NY, NZ, off2)                                                                                            \
    /* this step (the reads past a range's end stay inside the padded planes and are masked by `left`) */               \
    const int left = ecode - code; /* 0 for a finished thread */                                                         \
    const f32x2 xa = {X.x, X.y}, xb = {X.z, X.w}, ya = {Y.x, Y.y}, yb = {Y.z, Y.w}, za = {Z.x, Z.y}, zb = {Z.z, Z.w};    \
    /* nanoflann.hpp:432-440: result += diff*diff for x, y, z starting from 0 (two candidates per op) */                 \
    const f32x2 dxa = tq_pk_sub(qx, xa), dya = tq_pk_sub(qy, ya), dza = tq_pk_sub(qz, za);                               \
    const f32x2 dxb = tq_pk_sub(qx, xb), dyb = tq_pk_sub(qy, yb), dzb = tq_pk_sub(qz, zb);                               \
    const f32x2 da = (dxa * dxa + dya * dya) + dza * dza;                                                                \
    const f32x2 db = (dxb * dxb + dyb * dyb) + dzb * dzb;                                                                \
    const unsigned dbits[4] = {__float_as_uint(da.x), __float_as_uint(da.y), __float_as_uint(db.x), __float_as_uint(db.y)}; \
    const f32x2 fa = da * scale, fb = db * scale;                                                                        \
    const float fd[4] = {fa.x, fa.y, fb.x, fb.y};                                                                        \
    if (PASS != 1) n = min(n, NET + 1);                                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                      \
      bool h = u < left && dbits[u] < r2b; /* both are non-negative floats; NaN sorts above everything */               \
      if (PASS == 2) h = h && fd[u] < fcut;                                                                              \
      if (PASS == 1) {                                                                                                   \
        const unsigned fx = min(__float2uint_rz(fd[u]), FIX_MAX);                                                        \
        __hip_atomic_fetch_add(&hist[(fx >> (FB - 5)) * WAVE + lane], h ? 1u << ((fx >> (FB - 10)) & 16u) : 0u,          \
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);                                          \
      } else {                                                                                                           \
        my[n * WAVE] = (unsigned short)(code + u);                                                                       \
      }                                                                                                                  \
      n += h ? 1 : 0;                                                                                                    \
    }                                                                                                                    \
    code = c2, ecode = e2, badj = b2;                                                                                    \
  }
    tq_f4u X0, Y0, Z0, X1, Y1, Z1;
    {
      const unsigned off = code < ecode ? (unsigned)(badj + code) << 2 : 0u;
      TQ_LOAD(X0, Y0, Z0, off)
    }
    while (__any(code < ecode)) {
      TQ_STEP(X0, Y0, Z0, X1, Y1, Z1)
      TQ_STEP(X1, Y1, Z1, X0, Y0, Z0)
    }
#undef TQ_STEP
#undef TQ_LOAD
  };
    if (PRESEL) {
      unsigned int* hist = reinterpret_cast<unsigned int*>(lists);
#pragma unroll
      for (int i = 0; i < 32; ++i) hist[i * WAVE + lane] = 0u;
      scan(std::integral_constant<int, 1>{});
      ntrue = n;
      // first bin at which the running count reaches the row width (lanes with <= NET hits list everything)
      const int need = min(width, NET);
      int cum = 0, bsel = 64;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const unsigned v = hist[i * WAVE + lane];
        cum += (int)(v & 0xffffu);
        bsel = (bsel == 64 && cum >= need) ? 2 * i : bsel;
        cum += (int)(v >> 16);
        bsel = (bsel == 64 && cum >= need) ? 2 * i + 1 : bsel;
      }
      // (bin 63 holds the clamped fixed-point values: keeping it means keeping everything; rows wider than the network
      // keep everything too -- such a query is finished by the wave as in the plain kernel)
      fcut = (ntrue > NET && width <= NET && bsel < 63) ? (float)((unsigned)(bsel + 1) << (FB - 6)) : INFINITY;
      n = 0;
      scan(std::integral_constant<int, 2>{});
    } else {
      scan(std::integral_constant<int, 0>{});
      ntrue = n;
    }
  }
  TQ_STOP(2, n)
  const bool big = n > NET;
  // a wave nearly all of whose queries are beyond the network is not what this kernel is for: give up before the sort (the
  // caller repeats the call on the next kernel and remembers the site).  Fewer are finished here, one after the other.
  if (__popcll(__ballot(big)) > TQ_HOPELESS) blk_flag = 3;
  const int m = min(n, NET);
  const int wmax_u = __builtin_amdgcn_readfirstlane(wave_max_i32_dpp(m));
  // ---- keys: list entries -> one word per hit; slots past the hit count hold pad words (distinct distance fields above
  //      every real one, so pads never look like ties)
  unsigned key[NET];
  // The support index of slot s goes to LDS as idx32[s][lane], IN PLACE over the lists: row s of the u16 lists is bytes
  // [128 s, 128 s + 128), idx32 row s is bytes [256 s, 256 s + 256) -- walking the slots downwards, a group's idx rows only
  // cover list rows that this or an earlier group has already read.
  unsigned int* idx32 = reinterpret_cast<unsigned int*>(&lds.s);
#pragma unroll
  for (int s8 = NET - 8; s8 >= 0; s8 -= 8) {
    if (s8 < wmax_u) {  // (uniform) eight hits at a time: one 16-byte gather each from the cell-sorted records
      float4 sp[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = (int)lists[(s8 + u) * WAVE + lane];
        sp[u] = sorted_s[s8 + u < m ? lbase[min(c >> 12, NBAND - 1) * WAVE + lane] + c : 0];
      }
      __syncthreads();  // (the list reads above are done before the rows are overwritten)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float dx = qp.x - sp[u].x, dy = qp.y - sp[u].y, dz = qp.z - sp[u].z;
        const float d = (dx * dx + dy * dy) + dz * dz;  // same arithmetic as the test: same bits
        const unsigned fix = min(__float2uint_rz(d * scale), FIX_MAX);
        key[s8 + u] = s8 + u < m ? (fix << SB) | (unsigned)(s8 + u) : ((((1u << FB) - (unsigned)NET + (unsigned)(s8 + u)) << SB) | (unsigned)(s8 + u));
        if (NET <= 32) idx32[(s8 + u) * WAVE + lane] = __float_as_uint(sp[u].w);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) key[s8 + u] = (((1u << FB) - (unsigned)NET + (unsigned)(s8 + u)) << SB) | (unsigned)(s8 + u);
    }
  }
  __syncthreads();
  TQ_STOP(3, (int)(key[0] ^ key[13] ^ key[31]))
  bool tie = false;
  if (wmax_u > 1) {
#pragma unroll
    for (int c = 0; c < tq_network<NET>.n; ++c) {
      const unsigned ka = key[tq_network<NET>.a[c]], kb = key[tq_network<NET>.b[c]];
      key[tq_network<NET>.a[c]] = min(ka, kb);
      key[tq_network<NET>.b[c]] = max(ka, kb);
    }
    unsigned tmin = 0xffffffffu;
#pragma unroll
    for (int i = 0; i + 1 < NET; ++i) tmin = min(tmin, key[i] ^ key[i + 1]);
    tie = tmin < (unsigned)NET;  // two neighbours with the same distance field
  }
  TQ_STOP(4, (int)(key[0] ^ key[13] ^ key[31]) + (tie ? 1 : 0))
  // ---- rows of the queries the network finished
  const bool slow = valid && !blk_flag && (big || tie);
  int hmax = 0, nslow = 0;
  if (!blk_flag) {
    // sorted list slots -> support indices (in the key registers); then idx32 is dead and the rows are transposed
    // through LDS in blocks of 16 columns (a thread storing its own row is 64 rows per store instruction: measured
    // 0.17 ms for 8 x 200 k x 40)
#pragma unroll
    for (int i = 0; i < NET; ++i)
      if (i < wmax_u) {  // (uniform guard)
        if (NET <= 32) {
          key[i] = idx32[(int)(key[i] & (unsigned)(NET - 1)) * WAVE + lane];
        } else {
          const int c = (int)lists[(int)(key[i] & (unsigned)(NET - 1)) * WAVE + lane];
          key[i] = __float_as_uint(sorted_s[i < m ? lbase[min(c >> 12, NBAND - 1) * WAVE + lane] + c : 0].w);
        }
      }
    TQ_STOP(5, (int)(key[0] ^ key[13] ^ key[31]))
    __syncthreads();
    lds.r.qinfo[lane] = make_int2(orig, (valid && !slow) ? m : -1);
    const int rows_here = min(WAVE, nq - blk * WAVE);
    if (DIRECT) {
      // int64 rows in the caller's (original) order: eight lanes write the 128 bytes a row has in the block, eight rows per
      // store instruction.  Streaming stores when every piece covers whole 64-byte sectors (width % 8 == 0: measured
      // 0.33 -> 0.27 ms, the rows stop evicting the candidate planes from the L2s); on rows that do not start on a sector
      // they cost partial-sector writes instead (measured 0.29 -> 0.45 ms on width 46), so those go through the L2s
      // (two instantiations: inside one loop the compiler merges the two stores into a plain one)
      auto write_blocks = [&](auto stream_tag) {
        constexpr bool STREAM = decltype(stream_tag)::value;
#pragma unroll
      for (int cb = 0; cb < NET; cb += L::RB) {
        if (cb < width) {  // (uniform)
          if (cb < wmax_u) {
#pragma unroll
            for (int i = 0; i < L::RB; ++i) lds.r.rowbuf[lane * L::RS + i] = key[cb + i];
          }
          __syncthreads();
#pragma unroll 2
          for (int e = lane; e < WAVE * (L::RB / 2); e += WAVE) {
            const int r = e >> 3, ci = (e & 7) * 2, cc = cb + ci;
            const int2 qi = lds.r.qinfo[r];
            const unsigned vx = lds.r.rowbuf[r * L::RS + ci];
            const unsigned vy = lds.r.rowbuf[r * L::RS + ci + 1];
            if (r < rows_here && qi.y >= 0 && cc < width) {  // (qi.y < 0: the wave finishes this row below)
              int64_t* dst = out + (int64_t)qi.x * width + cc;
              const long long ox = cc < qi.y ? (long long)vx : (long long)pad_value;
              const long long oy = cc + 1 < qi.y ? (long long)vy : (long long)pad_value;
              if (cc + 1 < width) {
                tq_ll2 o;
                o.x = ox;
                o.y = oy;
                if (STREAM) __builtin_nontemporal_store(o, reinterpret_cast<tq_ll2*>(dst));
                else *reinterpret_cast<tq_ll2*>(dst) = o;
              } else {
                dst[0] = ox;
              }
            }
          }
          __syncthreads();
        }
      }
      };
      if ((width & 7) == 0) write_blocks(std::true_type{});
      else write_blocks(std::false_type{});
      // columns past the network's reach are padding for every row the network finished
      for (int cb = NET; cb < width; cb += L::RB) {
        for (int e = lane; e < WAVE * (L::RB / 2); e += WAVE) {
          const int r = e >> 3, cc = cb + (e & 7) * 2;
          const int2 qi = lds.r.qinfo[r];
          if (r < rows_here && qi.y >= 0 && cc < width) {
            int64_t* dst = out + (int64_t)qi.x * width + cc;
            if (cc + 1 < width) {
              tq_ll2 o;
              o.x = (long long)pad_value;
              o.y = (long long)pad_value;
              *reinterpret_cast<tq_ll2*>(dst) = o;
            } else {
              dst[0] = pad_value;
            }
          }
        }
      }
      __syncthreads();
    } else {
      // compact rows: u32, two halves of TQ_ROW_HALF per query, in the caller's (ORIGINAL) order -- the scatter happens here,
      // as whole 64-byte sectors (four lanes write the 64 bytes a row has in the block), so that tq_expand_kernel is two
      // sequential streams; the counts go with them.
      if (valid && !slow) q_cnt[orig] = n;  // (a wave-finished query's count is written below)
      static_assert(NET <= TQ_ROW_CAP, "the network's rows fit the two halves");
#pragma unroll
      for (int cb = 0; cb < NET; cb += L::RB) {
        if (cb < wmax_u) {  // (uniform) entries past a query's own count are never read
#pragma unroll
          for (int i = 0; i < L::RB; ++i) lds.r.rowbuf[lane * L::RS + i] = key[cb + i];
          __syncthreads();
#pragma unroll
          for (int e = lane; e < WAVE * (L::RB / 4); e += WAVE) {
            const int r = e >> 2, ci = (e & 3) * 4;
            uint4 v;
            v.x = lds.r.rowbuf[r * L::RS + ci];
            v.y = lds.r.rowbuf[r * L::RS + ci + 1];
            v.z = lds.r.rowbuf[r * L::RS + ci + 2];
            v.w = lds.r.rowbuf[r * L::RS + ci + 3];
            const int2 qi = lds.r.qinfo[r];
            if (qi.y >= 0)
              *reinterpret_cast<uint4*>(rows32 + (cb < TQ_ROW_HALF ? (size_t)0 : rows_hi) + (size_t)qi.x * TQ_ROW_HALF +
                                        (cb & (TQ_ROW_HALF - 1)) + ci) = v;
          }
          __syncthreads();
        }
      }
    }
    // ---- wave-finished queries, one after the other: lanes = candidates, hits compacted by ballot, ranked by counting
    unsigned long long slowm = __ballot(slow);
    nslow = __popcll(slowm);  // (reported: a call with many of these tells the caller to start the site on another kernel)
    while (slowm) {
      const int sl = __ffsll((long long)slowm) - 1;
      slowm &= slowm - 1ull;
      const float bx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qp.x), sl));
      const float by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qp.y), sl));
      const float bz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qp.z), sl));
      const int borig = __builtin_amdgcn_readlane(orig, sl);
      const float bcut = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fcut), sl));  // (PRESEL; else +inf)
      const int btrue = __builtin_amdgcn_readlane(ntrue, sl);
      int s0[NBAND], len[NBAND];
      int tot = 0;
#pragma unroll
      for (int k = 0; k < NBAND; ++k) {
        s0[k] = __builtin_amdgcn_readlane(p0[k], sl);
        len[k] = __builtin_amdgcn_readlane(p1[k], sl) - s0[k];
        tot += len[k];
      }
      int h = 0;  // uniform
      for (int c0 = 0; c0 < tot; c0 += WAVE) {
        const int c = c0 + lane;
        const bool in = c < tot;
        int p = 0;
        {
          int rem = in ? c : 0;
          bool found = false;
#pragma unroll
          for (int k = 0; k < NBAND; ++k) {
            if (!found && rem < len[k]) {
              p = s0[k] + rem;
              found = true;
            }
            rem -= found ? 0 : len[k];
          }
        }
        const float dx = bx - px[p], dy = by - py[p], dz = bz - pz[p];
        const float d = (dx * dx + dy * dy) + dz * dz;
        const bool hit = in && __float_as_uint(d) < r2b && (!PRESEL || d * scale < bcut);
        const unsigned long long hm = __ballot(hit);
        if (hit) {
          const int pos = h + __popcll(hm & ((1ull << lane) - 1ull));
          if (pos < TQ_BKEYS) lds.bkeys[pos] = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(sorted_s[p].w);
        }
        h += __popcll(hm);
      }
      if (h > TQ_BKEYS) {  // (uniform)
        blk_flag = 5;
        break;
      }
      hmax = max(hmax, PRESEL ? btrue : h);
      __syncthreads();
      for (int e = lane; e < h; e += WAVE) {
        const unsigned long long ke = lds.bkeys[e];
        int rank = 0;
        int j2 = 0;
        for (; j2 + 8 <= h; j2 += 8) {  // eight independent (broadcast) reads in flight
          unsigned long long kj[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) kj[u] = lds.bkeys[j2 + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) rank += kj[u] < ke ? 1 : 0;
        }
        for (; j2 < h; ++j2) rank += lds.bkeys[j2] < ke ? 1 : 0;
        if (DIRECT) {
          if (rank < width) out[(int64_t)borig * width + rank] = (int64_t)(unsigned int)(ke & 0xffffffffull);
        } else {
          if (rank < TQ_ROW_CAP)
            rows32[(rank < TQ_ROW_HALF ? (size_t)0 : rows_hi) + (size_t)borig * TQ_ROW_HALF + (rank & (TQ_ROW_HALF - 1))] =
                (uint32_t)(ke & 0xffffffffull);
        }
      }
      if (DIRECT) {
        for (int c = h + lane; c < width; c += WAVE) out[(int64_t)borig * width + c] = pad_value;
      } else if (lane == 0) {
        q_cnt[borig] = h;
      }
      __syncthreads();
    }
  }
  const int nmax = __builtin_amdgcn_readfirstlane(wave_max_i32_dpp(slow ? 0 : ntrue));
  if (lane == 0) {
    blk_stats[2 * blk] = max(nmax, hmax);
    blk_stats[2 * blk + 1] = blk_flag | (nslow << 8);
  }
}

// Compact rows -> int64 rows of the final width (the bare radius_neighbors: the width is the largest count, known to the
// host between the two launches).  Both sides are in the caller's row order (tq_kernel scatters its compact rows there as
// whole sectors), so this is a widening copy of two sequential streams: thread = (row, group of four columns).
// (Compact rows in cell order + a gather here measured 0.22 ms for 205 + 589 MB, this form 0.18 - 0.20 ms.)
__global__ __launch_bounds__(256) void tq_expand_kernel(const uint32_t* __restrict__ rows32, const int32_t* __restrict__ q_cnt,
                                                        size_t rows_hi, int nq, int width,
                                                        int64_t pad_value, int64_t* __restrict__ out) {
  const int groups = (width + 3) >> 2;
  const unsigned magic = 0xffffffffu / (unsigned)groups + 1u;  // (e / groups for e < 2^16 only: rows are split per block)
  // a block owns 256 / groups whole rows... keep it simple: 64 rows per block, threads loop over (row, group)
  const int row0 = blockIdx.x * 64;
  __shared__ int2 info[64];  // (row, count)
  if (threadIdx.x < 64) {
    const int o = row0 + threadIdx.x;
    int2 v = make_int2(0, 0);
    if (o < nq) {
      v.x = o;
      v.y = min(q_cnt[o], width);
    }
    info[threadIdx.x] = v;
  }
  __syncthreads();
  const int rows_here = min(64, nq - row0);
  const int total = rows_here * groups;
  (void)magic;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int r = e / groups, g = e - r * groups, cc = g * 4;
    const int2 qi = info[r];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (cc < qi.y)
      v = *reinterpret_cast<const uint4*>(rows32 + (cc < TQ_ROW_HALF ? (size_t)0 : rows_hi) + (size_t)qi.x * TQ_ROW_HALF + (cc & (TQ_ROW_HALF - 1)));
    int64_t* dst = out + (int64_t)(row0 + r) * width + cc;
    const long long o0 = cc < qi.y ? (long long)v.x : (long long)pad_value;
    const long long o1 = cc + 1 < qi.y ? (long long)v.y : (long long)pad_value;
    const long long o2 = cc + 2 < qi.y ? (long long)v.z : (long long)pad_value;
    const long long o3 = cc + 3 < qi.y ? (long long)v.w : (long long)pad_value;
    if (cc + 3 < width) {
      tq_ll2 a, b;
      a.x = o0;
      a.y = o1;
      b.x = o2;
      b.y = o3;
      __builtin_nontemporal_store(a, reinterpret_cast<tq_ll2*>(dst));  // one sequential stream: whole sectors
      __builtin_nontemporal_store(b, reinterpret_cast<tq_ll2*>(dst + 2));
    } else {
      dst[0] = o0;
      if (cc + 1 < width) dst[1] = o1;
      if (cc + 2 < width) dst[2] = o2;
    }
  }
}
