// 3D-Gaussian-splatting rasterizer forward for MI355X (gfx950), batched over views.
//
// Interface replaced: diff_gaussian_rasterization.GaussianRasterizer.forward (public upstream
// graphdeco-inria/diff-gaussian-rasterization; NOT in the GaussReg tree).
// Numerics contract: oracle/rasterizer_oracle.c -- every fp32 operation below is written in the
// same order (fmaf where the oracle says fmaf, separate mul/add elsewhere; this TU is compiled
// with -ffp-contract=off), so images and radii are compared bit-exact with the oracle.
//
// Pipeline per batch of V views over one set of P Gaussians:
//   preprocess  1 thread / Gaussian, loops over the V cameras (inputs read once per batch;
//               cov3D built once): cull, project, cov2D, conic, radius, tile rect, SH -> RGB
//   depth sort  stable radix sort of (view, depth bits) -> per-view front-to-back Gaussian order
//   tile bin    NO second sort and no (tile, id) key stream: chunks of 2048 depth-ordered Gaussians are binned by tile inside
//               LDS and written out as one contiguous block per chunk (see "tile binning" below); a tile's list is the
//               concatenation of its per-chunk segments -- in depth order, i.e. exactly upstream's single sort of
//               (tile << 32 | depth) keys
//   blend       1 workgroup / tile (16x16 px, 4 waves): walks the tile's segments chunk by chunk, Gaussian parameters
//               staged through LDS in batches of 256, front-to-back alpha blending, deterministic exp
#include <algorithm>
#include <type_traits>
#include <vector>

#include <atomic>

#include "common.hpp"

namespace gr {
namespace {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;
constexpr int MAX_VIEWS = 64;  // cameras per preprocess launch (constant-memory table)

struct DevView {
  float view[16];
  float proj[16];
  float campos[3];
  float tanx, tany;
  float fx, fy;
  float scale_mod;
  float bg[3];
};

__device__ __constant__ float SH_C0 = 0.28209479177387814f;
__device__ __constant__ float SH_C1 = 0.4886025119029199f;
__device__ __constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                                          0.31539156525252005f, -1.0925484305920792f,
                                          0.5462742152960396f};
__device__ __constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                          -0.4570457994644658f, 0.3731763325901154f,
                                          -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

// Deterministic expf (x <= 0): identical operation sequence to oracle_exp_det().
__device__ __forceinline__ float exp_det(float x) {
  x = fmaxf(x, -86.0f);
  const float L2E = 1.44269504088896341f, MAGIC = 12582912.0f;
  const float t = x * L2E;
  const float tm = t + MAGIC;
  const float nf = tm - MAGIC;
  const float f = fmaf(x, L2E, -nf);
  float p = 1.3264815788716078e-3f;
  p = fmaf(p, f, 9.671512059867382e-3f);
  p = fmaf(p, f, 5.550733581185341e-2f);
  p = fmaf(p, f, 2.4022242426872253e-1f);
  p = fmaf(p, f, 6.931470036506653e-1f);
  p = fmaf(p, f, 1.0f);
  return __uint_as_float(__float_as_uint(p) + (__float_as_uint(tm) << 23));
}

__device__ __forceinline__ void xform4x3(const float* M, const float* p, float* o) {
  o[0] = fmaf(M[0], p[0], fmaf(M[4], p[1], fmaf(M[8], p[2], M[12])));
  o[1] = fmaf(M[1], p[0], fmaf(M[5], p[1], fmaf(M[9], p[2], M[13])));
  o[2] = fmaf(M[2], p[0], fmaf(M[6], p[1], fmaf(M[10], p[2], M[14])));
}
__device__ __forceinline__ void xform4x4(const float* M, const float* p, float* o) {
  o[0] = fmaf(M[0], p[0], fmaf(M[4], p[1], fmaf(M[8], p[2], M[12])));
  o[1] = fmaf(M[1], p[0], fmaf(M[5], p[1], fmaf(M[9], p[2], M[13])));
  o[2] = fmaf(M[2], p[0], fmaf(M[6], p[1], fmaf(M[10], p[2], M[14])));
  o[3] = fmaf(M[3], p[0], fmaf(M[7], p[1], fmaf(M[11], p[2], M[15])));
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float* sc, float mod, const float* q,
                                                     float* c6) {
  const float s0 = mod * sc[0], s1 = mod * sc[1], s2 = mod * sc[2];
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  float R[3][3];
  R[0][0] = 1.f - 2.f * (y * y + z * z);
  R[0][1] = 2.f * (x * y - r * z);
  R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);
  R[1][1] = 1.f - 2.f * (x * x + z * z);
  R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);
  R[2][1] = 2.f * (y * z + r * x);
  R[2][2] = 1.f - 2.f * (x * x + y * y);
  float M[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    M[0][i] = s0 * R[i][0];
    M[1][i] = s1 * R[i][1];
    M[2][i] = s2 * R[i][2];
  }
#define GR_SIG(i, j) fmaf(M[0][i], M[0][j], fmaf(M[1][i], M[1][j], M[2][i] * M[2][j]))
  c6[0] = GR_SIG(0, 0);
  c6[1] = GR_SIG(0, 1);
  c6[2] = GR_SIG(0, 2);
  c6[3] = GR_SIG(1, 1);
  c6[4] = GR_SIG(1, 2);
  c6[5] = GR_SIG(2, 2);
#undef GR_SIG
}

__device__ __forceinline__ void cov2d(const float* t_in, float fx, float fy, float tanx, float tany,
                                      const float* c6, const float* V, float* out3) {
  float t[3] = {t_in[0], t_in[1], t_in[2]};
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t[0] / t[2], tytz = t[1] / t[2];
  t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
  t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
  const float J00 = fx / t[2];
  const float J02 = -(fx * t[0]) / (t[2] * t[2]);
  const float J11 = fy / t[2];
  const float J12 = -(fy * t[1]) / (t[2] * t[2]);
  float A0[3], A1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    A0[j] = fmaf(J00, V[j * 4 + 0], J02 * V[j * 4 + 2]);
    A1[j] = fmaf(J11, V[j * 4 + 1], J12 * V[j * 4 + 2]);
  }
  const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float B0[3], B1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    B0[k] = fmaf(S[k][0], A0[0], fmaf(S[k][1], A0[1], S[k][2] * A0[2]));
    B1[k] = fmaf(S[k][0], A1[0], fmaf(S[k][1], A1[1], S[k][2] * A1[2]));
  }
  out3[0] = fmaf(A0[0], B0[0], fmaf(A0[1], B0[1], A0[2] * B0[2])) + 0.3f;
  out3[1] = fmaf(A1[0], B0[0], fmaf(A1[1], B0[1], A1[2] * B0[2]));
  out3[2] = fmaf(A1[0], B1[0], fmaf(A1[1], B1[1], A1[2] * B1[2])) + 0.3f;
}

// sh: this Gaussian's coefficients, (M,3) row-major, already in registers/local memory
template <typename ShLoad>
__device__ __forceinline__ void sh_to_rgb(int deg, const float* pos, const float* campos,
                                          ShLoad S, float* rgb) {
  const float dx = pos[0] - campos[0], dy = pos[1] - campos[1], dz = pos[2] - campos[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx / len, y = dy / len, z = dz / len;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float res = SH_C0 * S(0, c);
    if (deg > 0) {
      res = res - SH_C1 * y * S(1, c) + SH_C1 * z * S(2, c) - SH_C1 * x * S(3, c);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * S(4, c) + SH_C2[1] * yz * S(5, c) +
              SH_C2[2] * (2.0f * zz - xx - yy) * S(6, c) + SH_C2[3] * xz * S(7, c) +
              SH_C2[4] * (xx - yy) * S(8, c);
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * S(9, c) + SH_C3[1] * xy * z * S(10, c) +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11, c) +
                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12, c) +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13, c) +
                SH_C3[5] * z * (xx - yy) * S(14, c) + SH_C3[6] * x * (xx - 3.0f * yy) * S(15, c);
        }
      }
    }
    res += 0.5f;
    rgb[c] = fmaxf(res, 0.0f);
  }
}

static_assert(TILE == RASTER_TILE, "get_rect / rect_decode (common.hpp) are written for this tile size");

// geometry state: one 64-byte record per (view, Gaussian) so that the blend's per-instance gather
// (ids arrive in depth order, i.e. random in memory) touches ONE 64-B sector instead of three lines:
//   rec[0] = {px, py, sxx, syy (axis cull factors)}   rec[3] = {radius(int bits), depth, -, -}   rec[1] = {conic.x, conic.y, conic.z, opacity}
//   rec[2] = {r, g, b, kc (cull factor)}          rec[3] = unused
struct Geom {
  float4* rec;           // [V][P][4]
  uint32_t* dfield;      // [V*P] depth-sort field: rebased depth bits (see KEY_DEPTH_BITS), 0 = culled
  uint32_t* rect_raw;    // [V*P] packed tile rectangle of every (view, Gaussian), Gaussian order
  uint64_t* keys_a;      // [V*P] x 2: (field << 32 | id) ping-pong buffers of the depth sort
  uint64_t* keys_b;
  int32_t* order_a;      // (unused: the sort generates ids on the fly); order_b = per-view front-to-back order
  int32_t* order_b;
  uint32_t* rects;       // [V*P] depth-ordered tile rectangles (26-bit packing of the key's high bits)
  uint16_t* chunk_cnt;   // [V][nchunk][tiles] instances per (chunk of BIN_CHUNK depth-ordered Gaussians, tile)
  uint32_t* seg_off;     // [V][nchunk][tiles + 1] position of segment (chunk, tile) in the point list (+ end sentinel)
  int32_t* chunk_total;  // [V][nchunk] instances per chunk, then [V][nchunk] exclusive prefix inside the view
  int32_t* chunk_max;    // [1] largest chunk total
  void* ds_table;        // depth-sort histograms / offsets
  size_t ds_table_bytes;
  int2* key_mm;          // [V][ceil(P / 256)] smallest / largest depth field of a preprocess block (a few views per call only)
  int32_t* nvis;         // [V] visible (depth-ordered) Gaussians per view
  int32_t* totals;       // [V] instances per view, [V] largest chunk total of every view, [1] depth-overflow flag, [3] pad --
  DevView* views;        // -- immediately followed by the [MAX_VIEWS] camera table: ONE upload clears the flag and sets the cameras
  int32_t* scan_ws;
  size_t bytes;
};

constexpr int BIN_T = 256;                          // threads per binning workgroup (4 waves)
constexpr int BIN_CHUNK = 2048;                     // depth-ordered Gaussians per chunk (count: a workgroup, scatter: a wave)

Geom carve_geom(void* p, int64_t P, int V, int64_t tiles) {
  Geom g;
  Carver c(p);
  const int64_t nchunk = (P + BIN_CHUNK - 1) / BIN_CHUNK;
  g.rec = c.take<float4>(P * V * 4);
  g.dfield = c.take<uint32_t>(P * V);
  g.rect_raw = c.take<uint32_t>(P * V);
  g.keys_a = c.take<uint64_t>(P * V);
  g.keys_b = c.take<uint64_t>(P * V);
  g.order_a = nullptr;
  g.order_b = c.take<int32_t>(P * V);
  g.rects = c.take<uint32_t>(P * V);
  g.chunk_cnt = c.take<uint16_t>(V * nchunk * tiles);
  g.seg_off = c.take<uint32_t>(V * nchunk * (tiles + 1));
  g.chunk_total = c.take<int32_t>(2 * V * nchunk);
  g.chunk_max = c.take<int32_t>(1);
  g.ds_table_bytes = depth_sort_table_bytes(P, V);
  g.ds_table = c.take<char>(g.ds_table_bytes);
  g.key_mm = c.take<int2>((V <= 4 ? V : 0) * ((P + 255) / 256));  // (depth_sort_msd_possible)
  g.nvis = c.take<int32_t>(V);
  g.totals = c.take<int32_t>(2 * V + 4 + MAX_VIEWS * (sizeof(DevView) / sizeof(int32_t)));
  g.views = reinterpret_cast<DevView*>(g.totals ? g.totals + 2 * V + 4 : nullptr);
  static_assert(sizeof(DevView) % sizeof(int32_t) == 0, "camera table follows an int array");
  g.scan_ws = c.take<int32_t>(V * scan_ws_ints(nchunk));
  g.bytes = c.used();
  return g;
}

struct Bin {
  int32_t* point_list;  // [R] Gaussian ids: chunk-major, tile-sorted inside a chunk, depth order inside a segment
  size_t bytes;
};

Bin carve_bin(void* p, int64_t R, int64_t vtiles) {
  Bin b;
  Carver c(p);
  (void)vtiles;
  b.point_list = c.take<int32_t>(R + 64);
  b.bytes = c.used();
  return b;
}

// ------------------------------------------------------------------------------------ preprocess
// Per (view, Gaussian) the preprocess emits a 32-bit depth-sort field and the packed tile rectangle (26 bits:
// x:7 | y:7 | w:6 | h:6; rectangles that do not fit store RECT_MARKER26 and are rebuilt from the record).  The sort moves
// (field << 32 | Gaussian id) words and looks the rectangle up by id in its last pass.
// Field: float bits of depth minus those of 0.125 (visible depths are > 0.2, so this is > 0 and order-preserving; < 2^27
// while depth < 8192): 27 bits sort in three 9-bit passes.  A depth >= 8192 raises a flag and the call re-sorts on the
// full 32 depth bits (slow path, never taken by sane scenes).
constexpr int KEY_DEPTH_BITS = 27;
constexpr uint32_t KEY_DEPTH_BASE = 0x3E000000u;  // bits of 0.125f

__device__ __forceinline__ uint32_t pack_rect(const int* rmin, const int* rmax) {
  const int w = rmax[0] - rmin[0], h = rmax[1] - rmin[1];
  if (w > 63 || h > 63 || rmin[0] > 126 || rmin[1] > 126) return RECT_MARKER26;
  return (uint32_t)rmin[0] | ((uint32_t)rmin[1] << 7) | ((uint32_t)w << 14) | ((uint32_t)h << 20);
}

constexpr int REC_PLANE = 66;  // float4 per record-piece plane (64 + 2): the transposed ds_read_b128 are conflict-free
constexpr int SH_ROW = 13;  // float4 per staged Gaussian: 12 used + 1 pad -> conflict-free ds_read_b128

// SH16: 16 coefficients per Gaussian, 16-byte aligned.  LATE (one view per call): the 192 bytes are fetched only by the
// Gaussians that survive the culling of THE view (12 float4 loads per visible lane, straight into registers) -- with a
// single camera 40 % of the benchmark scene never reads its SH; with several views every Gaussian is visible somewhere,
// and the coalesced stream through LDS below is the better way.
template <bool HAS_SH, bool HAS_COV, bool SH16, bool LATE>
__global__ __launch_bounds__(256) void preprocess_kernel(
    int P, int D, int M, int V, const DevView* __restrict__ views, const float* __restrict__ means3D,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ opacities, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, int W, int H,
    int32_t* __restrict__ radii, float4* __restrict__ rec, uint32_t* __restrict__ dfield,
    uint32_t* __restrict__ rect_raw, int32_t* __restrict__ far_flag, DevView cam1, int far_seq,
    int2* __restrict__ key_mm) {
  // LATE (one camera per call): the camera arrives in the kernel arguments `cam1` (no upload in front of the frame); the first
  // block leaves it in `views` for the kernels behind this one.  far_seq: the value a far depth stores into *far_flag
  // (a per-call stamp when nobody cleared the flag, else 1).
  if (LATE && blockIdx.x == 0 && threadIdx.x < (int)(sizeof(DevView) / 4))
    reinterpret_cast<float*>(const_cast<DevView*>(views))[threadIdx.x] = reinterpret_cast<const float*>(&cam1)[threadIdx.x];
  __shared__ float4 s_sh[(SH16 && !LATE) ? WAVE * SH_ROW : 1];
  __shared__ float4 s_rec[256 / WAVE][4 * REC_PLANE];
  __shared__ int s_mm[2][2][256 / WAVE];  // [view parity][min, max][wave]
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float shr[SH16 ? 48 : 1];
  if (SH16 && !LATE) {
    // degree-3 SH = 192 B per Gaussian: the block's 48 KB are read as a coalesced float4 stream and transposed through
    // LDS (row stride 13 float4 keeps the per-thread ds_read_b128 conflict-free), one wave's 64 Gaussians at a time through
    // the same 13 KB -- a 52 KB staging area for all four waves capped the CU at 12 resident waves for the whole view loop.
    const int g0 = blockIdx.x * 256;
    const int n_here = min(256, P - g0);
    const float4* src = reinterpret_cast<const float4*>(shs + (int64_t)g0 * 48);
#pragma unroll 1
    for (int w = 0; w < 256 / WAVE; ++w) {
      const int lim = min(WAVE, n_here - w * WAVE) * 12;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int f = threadIdx.x + u * 256;
        if (f < lim) {
          const int g = f / 12, j = f - g * 12;
          s_sh[g * SH_ROW + j] = src[w * WAVE * 12 + f];
        }
      }
      __syncthreads();
      if ((int)threadIdx.x / WAVE == w && i < P) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const float4 t = s_sh[(threadIdx.x & (WAVE - 1)) * SH_ROW + j];
          shr[4 * j] = t.x;
          shr[4 * j + 1] = t.y;
          shr[4 * j + 2] = t.z;
          shr[4 * j + 3] = t.w;
        }
      }
      __syncthreads();
    }
  }
  // Threads past the end stay alive (they help to write their wave's records below) on a clamped index and store nothing.
  const bool valid = i < P;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave_first = i - lane;  // first Gaussian of this wave
  i = min(i, P - 1);
  const float p[3] = {means3D[3 * (int64_t)i], means3D[3 * (int64_t)i + 1], means3D[3 * (int64_t)i + 2]};
  const float opacity = opacities[i];
  float c6[6];
  float sc[3], rot[4];
  if (HAS_COV) {
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (int64_t)i + k];
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) sc[k] = scales[3 * (int64_t)i + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) rot[k] = rotations[4 * (int64_t)i + k];
  }
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const float* sh = HAS_SH ? shs + (int64_t)i * M * 3 : nullptr;
  float cpre[3] = {0.f, 0.f, 0.f};
  if (!HAS_SH) {
#pragma unroll
    for (int k = 0; k < 3; ++k) cpre[k] = colors_precomp[3 * (int64_t)i + k];
  }
  float mod_prev = 0.f;
  bool have_cov = HAS_COV;
  for (int v = 0; v < V; ++v) {
    const DevView& cam = LATE ? cam1 : views[v];
    const int64_t o = (int64_t)v * P + i;
    int out_radius = 0;
    uint32_t out_field = 0u, out_rect = 0u;  // culled: depth field 0
    float out_depth = 0.f, out_sxx = INFINITY, out_syy = INFINITY;
    float2 out_xy = make_float2(0.f, 0.f);
    float4 out_co = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 out_rgb = make_float4(0.f, 0.f, 0.f, 0.f);
    float pv[3];
    xform4x3(cam.view, p, pv);
    if (pv[2] > 0.2f) {
      float ph[4];
      xform4x4(cam.proj, p, ph);
      const float pw = 1.0f / (ph[3] + 0.0000001f);
      const float pprojx = ph[0] * pw, pprojy = ph[1] * pw;
      if (!HAS_COV && (!have_cov || cam.scale_mod != mod_prev)) {
        cov3d_from_scale_rot(sc, cam.scale_mod, rot, c6);
        have_cov = true;
        mod_prev = cam.scale_mod;
      }
      float cv[3];
      cov2d(pv, cam.fx, cam.fy, cam.tanx, cam.tany, c6, cam.view, cv);
      const float det = cv[0] * cv[2] - cv[1] * cv[1];
      if (det != 0.0f) {
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (cv[0] + cv[2]);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l1 = mid + sq, l2 = mid - sq;
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        const float px = ((pprojx + 1.0f) * (float)W - 1.0f) * 0.5f;
        const float py = ((pprojy + 1.0f) * (float)H - 1.0f) * 0.5f;
        int rmin[2], rmax[2];
        get_rect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        const int ntile = (rmax[0] - rmin[0]) * (rmax[1] - rmin[1]);
        if (ntile != 0) {
          float rgb[3];
          if (HAS_SH) {
            if (SH16 && LATE) {
              const float4* s4 = reinterpret_cast<const float4*>(shs + (int64_t)i * 48);
#pragma unroll
              for (int jj = 0; jj < 12; ++jj) {
                const float4 t4 = s4[jj];
                shr[4 * jj] = t4.x;
                shr[4 * jj + 1] = t4.y;
                shr[4 * jj + 2] = t4.z;
                shr[4 * jj + 3] = t4.w;
              }
            }
            if (SH16) sh_to_rgb(D, p, cam.campos, [&](int k, int c) { return shr[k * 3 + c]; }, rgb);
            else sh_to_rgb(D, p, cam.campos, [&](int k, int c) { return sh[k * 3 + c]; }, rgb);
          } else {
            rgb[0] = cpre[0]; rgb[1] = cpre[1]; rgb[2] = cpre[2];
          }
          out_depth = pv[2];
          out_radius = (int)my_radius;
          out_xy = make_float2(px, py);
          out_co = make_float4(cv[2] * det_inv, -cv[1] * det_inv, cv[0] * det_inv, opacity);
          // rgb.w: k such that |d|^2 > |pc| * k  ==>  fp32 power < pc for any cutoff pc < 0 (blend
          // cell culling).  power <= -|d|^2 (0.5/l1 - 2e-6): the 2e-6 covers the fp32 evaluation
          // error of the quadratic form given lambda_min(cov) >= 0.3 (the +0.3 dilation).
          const bool cullable = det > 0.0f && l2 >= 0.29f && l1 < 1.0e4f;
          const float kc = cullable ? 1.001f / (0.5f / l1 - 2.0e-6f) : INFINITY;
          out_rgb = make_float4(rgb[0], rgb[1], rgb[2], kc);
          // Tile rectangle actually emitted: the reference square (radius = ceil(3 sigma_max)) intersected with the
          // bounding box of the region where alpha can reach 1/255.  A pixel contributes only if fp32 power >= pc
          // (pc as in the blend, with margin); inside the cutoff circle rc2 the fp32 quadratic form is within
          // 2e-6 rc2 of the exact one, whose level set {0.5 d^T A d <= c'} has half-extents sqrt(2 c' Sigma_xx / yy).
          // Only (tile, Gaussian) pairs that blend nothing are dropped, so the image is unchanged; `radii` is not.
          if (cullable) {
            // blend cell culling along the axes: |dx|^2 > c' * sxx (or |dy|^2 > c' * syy) ==> no contribution
            out_sxx = 2.0f * cv[0] * 1.004f;
            out_syy = 2.0f * cv[2] * 1.004f;
            const float pcm = __logf(255.0f * opacity) + 2.0e-3f;
            if (pcm > 0.0f) {
              const float cp = pcm + 2.0e-6f * (pcm * kc);
              const float hx = sqrtf(2.0f * cp * cv[0]) * 1.001f + 1.0e-2f;
              const float hy = sqrtf(2.0f * cp * cv[2]) * 1.001f + 1.0e-2f;
              // pixels x with |x - px| <= hx: [ceil(px - hx), floor(px + hx)] -> tiles
              const float xlo = ceilf(px - hx), xhi = floorf(px + hx), ylo = ceilf(py - hy), yhi = floorf(py + hy);
              if (xlo > -1.0e6f && xhi < 1.0e6f && ylo > -1.0e6f && yhi < 1.0e6f) {
                rmin[0] = max(rmin[0], (int)floorf(xlo / (float)TILE));
                rmin[1] = max(rmin[1], (int)floorf(ylo / (float)TILE));
                rmax[0] = min(rmax[0], (int)floorf(xhi / (float)TILE) + 1);
                rmax[1] = min(rmax[1], (int)floorf(yhi / (float)TILE) + 1);
                if (rmax[0] < rmin[0]) rmax[0] = rmin[0];
                if (rmax[1] < rmin[1]) rmax[1] = rmin[1];
              }
            }
          }
          uint32_t dk = __float_as_uint(out_depth) - KEY_DEPTH_BASE;  // out_depth > 0.2 > 0.125
          if (dk >= (1u << KEY_DEPTH_BITS)) {
            dk = (1u << KEY_DEPTH_BITS) - 1;
            if (LATE) *far_flag = far_seq; else atomicOr(far_flag, 1);
          }
          out_field = dk;
          out_rect = pack_rect(rmin, rmax);
        }
      }
    }
    if (valid) {
      __builtin_nontemporal_store(out_radius, radii + o);  // (an output nobody in the pipeline reads)
      dfield[o] = out_field;
      rect_raw[o] = out_rect;
    }
    if (key_mm != nullptr) {  // the block's key range of this view, for the bucket sort (depth_sort.hip)
      const uint32_t fld = valid ? out_field : 0u;
      const int mn = wave_min_i32_dpp(fld != 0u ? (int)fld : 0x7fffffff), mx = wave_max_i32_dpp((int)fld);
      if (lane == 0) s_mm[v & 1][0][threadIdx.x / WAVE] = mn, s_mm[v & 1][1][threadIdx.x / WAVE] = mx;
      __syncthreads();  // (one barrier per view: the other half of s_mm is the one the next view writes)
      if (threadIdx.x == 0) {
        int bmn = 0x7fffffff, bmx = 0;
#pragma unroll
        for (int w = 0; w < 256 / WAVE; ++w) bmn = min(bmn, s_mm[v & 1][0][w]), bmx = max(bmx, s_mm[v & 1][1][w]);
        key_mm[(int64_t)v * gridDim.x + blockIdx.x] = make_int2(bmn, bmx);
      }
    }
    // The wave's 64 records (4 KB, contiguous) leave through LDS: lane l stores piece l % 4 of record 16 k + l / 4 in
    // store k, so every store instruction covers whole lines.  (Each lane writing its own record piece by piece costs four
    // partial-line writes per record: measured 0.26 ms of the 0.77 ms kernel at 32 views.)  Culled Gaussians are never
    // gathered: their 64-B line is not touched at all.
    const unsigned long long vis = __ballot(valid && out_radius > 0);
    float4* wrec = s_rec[threadIdx.x / WAVE];
    wrec[0 * REC_PLANE + lane] = make_float4(out_xy.x, out_xy.y, out_sxx, out_syy);
    wrec[1 * REC_PLANE + lane] = out_co;
    wrec[2 * REC_PLANE + lane] = out_rgb;
    wrec[3 * REC_PLANE + lane] = make_float4(__int_as_float(out_radius), out_depth, 0.f, 0.f);  // wide-rectangle fallback only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4* wout = rec + 4 * ((int64_t)v * P + wave_first);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rl = 16 * k + lane / 4;
      const float4 piece = wrec[(lane & 3) * REC_PLANE + rl];
      // streaming stores (whole 64-byte records, 1 KB per instruction): the records are next read by the blend, three
      // stages later and in another order -- kept out of the caches they no longer evict the depth fields and rectangles the
      // sort is about to read (32 views: preprocess 0.61 -> 0.56 ms, depth sort 0.46 -> 0.41 ms)
      if ((vis >> rl) & 1ull) {
        typedef float pre_f4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(pre_f4{piece.x, piece.y, piece.z, piece.w}, reinterpret_cast<pre_f4*>(wout + 4 * rl + (lane & 3)));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// slow path of the depth sort: the field becomes the full 32 depth bits (visible depths are > 0.2: non-zero)
__global__ __launch_bounds__(256) void full_keys_kernel(int64_t n, const float4* __restrict__ rec,
                                                        const int32_t* __restrict__ radii, uint32_t* __restrict__ dfield) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  dfield[o] = radii[o] > 0 ? __float_as_uint(rec[4 * o + 3].y) : 0u;
}

// ------------------------------------------------------------------------------------ tile binning
// Input: per view the Gaussians in depth order (ids) with their tile rectangles (26-bit packing, RECT_MARKER26 = "did not
// fit, rebuild from the record").  Output: point_list, CHUNK-major: the instances of one chunk of BIN_CHUNK consecutive
// depth-ordered Gaussians form one contiguous block, sorted by tile inside LDS (depth order kept inside every (chunk,
// tile) segment), and seg_off[view][chunk][tile] says where each segment starts.  The blend walks a tile's segments in
// chunk order = depth order, so it sees exactly what a stable sort of (tile, depth-rank) keys would give, but no key is
// ever materialised and every global write is a full coalesced line (scattering 4-byte ids straight into per-tile lists
// was measured first: 62 M partial-line writes per 32 views cost 1.1 ms in the L2 alone).
//   count    a workgroup owns a chunk, histograms its tiles in LDS -> chunk_cnt, chunk_total
//   scan     chunk totals -> chunk bases (per view); per chunk an exclusive scan over the tiles -> seg_off
//   scatter  a workgroup owns a chunk; wave w owns a quarter of its Gaussians and a private cursor per tile in LDS
//            (segment start + the earlier waves' counts).  64 Gaussians (lanes) per step:
//              small rectangles (<= 8 x 8 tiles, wave-uniform test): with M = 2 / 4 / 8 >= the largest side, a rectangle
//                holds at most ONE tile of every residue class (x mod M, y mod M), so the wave walks the M*M classes and
//                in each every lane issues the (at most one) tile it has there.  A given tile is therefore requested by
//                all its Gaussians in the SAME instruction, and LANE_ORDERED (one ds_add_rtn_u32; equal-address lanes
//                served in lane order -- probed on the device, see below) hands them consecutive slots in lane = depth
//                order.  Without that property the lanes of a tile find each other with one ballot per tile-id bit and
//                rank themselves explicitly.
//              a step containing a larger rectangle is walked one Gaussian at a time, lanes over its tiles.
//            A wave's LDS instructions execute in issue order, so every segment stays in depth order.  Ids land in an LDS
//            staging block at their final chunk-local position; the block is then copied out with coalesced stores
//            (chunks whose instances do not fit the staging block write straight to their final global position).
// With >= 8 views the workgroups of one view all run on the same XCD (block b sits on XCD b % 8).

// block -> (view, chunk); XCD-affine when there are at least 8 views
__device__ __forceinline__ bool bin_block(int V, int nchunk, int& v, int& c) {
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
inline int bin_grid(int V, int nchunk) { return V >= 8 ? 8 * ((V + 7) / 8) * nchunk : V * nchunk; }

__global__ __launch_bounds__(BIN_T) void tile_count_kernel(int P, int V, int gx, int gy, int nchunk,
                                                           const int32_t* __restrict__ nvis,
                                                           const uint32_t* __restrict__ rects,
                                                           const int32_t* __restrict__ ids,
                                                           const float4* __restrict__ rec,
                                                           uint16_t* __restrict__ chunk_cnt,
                                                           int32_t* __restrict__ chunk_total) {
  extern __shared__ unsigned int s_hist[];
  __shared__ int s_wsum[BIN_T / WAVE];
  int v, c;
  if (!bin_block(V, nchunk, v, c)) return;
  const int tiles = gx * gy;
  for (int T = threadIdx.x; T < tiles; T += BIN_T) s_hist[T] = 0u;
  __syncthreads();
  const int64_t vbase = (int64_t)v * P;
  const int nv = nvis[v];
  int mine = 0;
  // all of a thread's rectangles are requested before the first one is used: one memory round trip instead of eight
  uint32_t rr[BIN_CHUNK / BIN_T];
#pragma unroll
  for (int it = 0; it < BIN_CHUNK / BIN_T; ++it) {
    const int t = c * BIN_CHUNK + it * BIN_T + threadIdx.x;
    rr[it] = t < nv ? rects[vbase + t] : 0u;
  }
#pragma unroll
  for (int it = 0; it < BIN_CHUNK / BIN_T; ++it) {
    const int t = c * BIN_CHUNK + it * BIN_T + threadIdx.x;
    const uint32_t r = rr[it];
    if (r == 0u) continue;
    int x0, y0, w, h;
    if (!rect_decode(r, r == RECT_MARKER26 ? ids[vbase + t] : 0, vbase, rec, gx, gy, x0, y0, w, h)) continue;
    mine += w * h;
    for (int y = y0; y < y0 + h; ++y)
      for (int x = x0; x < x0 + w; ++x) atomicAdd(&s_hist[y * gx + x], 1u);
  }
  const int wsum = wave_sum_i32_dpp(mine);
  if ((threadIdx.x & (WAVE - 1)) == 0) s_wsum[threadIdx.x / WAVE] = wsum;
  __syncthreads();
  uint16_t* dst = chunk_cnt + ((int64_t)v * nchunk + c) * tiles;
  // a tile can appear at most once per Gaussian: counts <= BIN_CHUNK < 65536
  for (int T = threadIdx.x; T < tiles; T += BIN_T) dst[T] = (uint16_t)s_hist[T];
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int i = 0; i < BIN_T / WAVE; ++i) tot += s_wsum[i];
    chunk_total[v * nchunk + c] = tot;
  }
}

// largest chunk total (sizes the scatter's LDS staging block); one workgroup, no same-address atomics
__global__ __launch_bounds__(1024) void chunk_max_kernel(int n, const int32_t* __restrict__ chunk_total,
                                                         int32_t* __restrict__ chunk_max) {
  __shared__ int s_m[1024 / WAVE];
  int m = 0;
  for (int i = threadIdx.x; i < n; i += 1024) m = max(m, chunk_total[i]);
  m = wave_max_i32_dpp(m);
  if ((threadIdx.x & (WAVE - 1)) == 0) s_m[threadIdx.x / WAVE] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < 1024 / WAVE; ++i) m = max(m, s_m[i]);
    *chunk_max = m;
  }
}

// per chunk: seg_off[tile] = view base + chunk base + exclusive scan of the chunk's tile counts; [tiles] = end
// SELF (a few views per call): no scan launch ran before this kernel -- every workgroup sums the raw chunk totals in front
// of its own (at most a few thousand values), and the first chunk of every view leaves the view's total and its largest
// chunk total for the host.  One launch less in a frame that is a chain of launches.
template <bool SELF>
__global__ __launch_bounds__(256) void seg_scan_kernel(int V, int tiles, int nchunk, const uint16_t* __restrict__ chunk_cnt,
                                                       const int32_t* __restrict__ chunk_base /* per view; SELF: raw totals */,
                                                       int32_t* __restrict__ totals, uint32_t* __restrict__ seg_off,
                                                       int32_t* __restrict__ mail, int mail_seq) {
  // mail (SELF only; host-mapped pinned memory or null): [V] totals, [V] chunk maxima, [1] far flag word, [V] stamps -- the host
  // polls the stamps instead of waiting for a device-to-host copy behind an event (both sat in the stream in front of the
  // scatter: 4 us of copy + a 6 - 8 us hand-over gap per frame)
  __shared__ int s_w[256 / WAVE];
  __shared__ int s_carry;
  const int v = blockIdx.x / nchunk, c = blockIdx.x % nchunk;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  int vb = 0;
  if (SELF) {
    const int me = (int)blockIdx.x;  // = v * nchunk + c: everything in front belongs to earlier views / chunks
    int part = 0;
    for (int i = threadIdx.x; i < me; i += 256) part += chunk_base[i];
    part = wave_sum_i32_dpp(part);
    if (lane == 0) s_w[wv] = part;
    __syncthreads();
    vb = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    if (c == 0) {  // block-uniform
      int sum = 0, mx = 0;
      for (int i = threadIdx.x; i < nchunk; i += 256) {
        const int t = chunk_base[v * nchunk + i];
        sum += t;
        mx = max(mx, t);
      }
      sum = wave_sum_i32_dpp(sum);
      mx = wave_max_i32_dpp(mx);
      if (lane == 0) {
        s_w[wv] = sum;
      }
      __shared__ int s_m[256 / WAVE];
      if (lane == 0) s_m[wv] = mx;
      __syncthreads();
      if (threadIdx.x == 0) {
        const int tv = s_w[0] + s_w[1] + s_w[2] + s_w[3], mv = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
        totals[v] = tv;
        totals[V + v] = mv;
        if (mail != nullptr) {
          __hip_atomic_store(&mail[v], tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&mail[V + v], mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (v == 0) __hip_atomic_store(&mail[2 * V], totals[2 * V], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&mail[2 * V + 1 + v], mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      __syncthreads();
    }
  } else {
    for (int u = 0; u < v; ++u) vb += totals[u];
  }
  const uint16_t* src = chunk_cnt + ((int64_t)v * nchunk + c) * tiles;
  uint32_t* dst = seg_off + ((int64_t)v * nchunk + c) * (tiles + 1);
  if (threadIdx.x == 0) s_carry = SELF ? vb : vb + chunk_base[v * nchunk + c];
  __syncthreads();
  for (int T0 = 0; T0 < tiles; T0 += 256) {
    const int T = T0 + threadIdx.x;
    const int n = T < tiles ? (int)src[T] : 0;
    const int incl = wave_incl_scan_add_dpp(n);
    if (lane == WAVE - 1) s_w[wv] = incl;
    __syncthreads();
    int base = s_carry;
    for (int i = 0; i < wv; ++i) base += s_w[i];
    if (T < tiles) dst[T] = (uint32_t)(base + incl - n);
    __syncthreads();
    if (threadIdx.x == 255) s_carry = base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[tiles] = (uint32_t)s_carry;
}

// TT = threads per workgroup: 256 (four waves, each a quarter of the chunk) for many views per call; 1024 for a few views,
// where the launch has fewer workgroups than the chip has CUs and a workgroup's walk through its chunk IS the kernel's
// duration (16 waves, an eighth of the steps each: 33 -> 12 us per one-camera frame)
// SELF_SEG (a few views per call, behind the bucket depth sort): no count / scan launch ran before this one.  The chunk
// totals came out of the sort (chunk_total: raw, per (view, chunk)); every workgroup sums the ones in front of its own,
// scans the tile counts it has to build anyway and WRITES its row of seg_off for the blend; the first chunk of a view
// reports the view's total and largest chunk to the host (see seg_scan_kernel<true>, whose job this is otherwise).
struct SelfSeg {
  const int32_t* chunk_total;
  int32_t* totals;
  int32_t* mail;
  int mail_seq;
};
template <bool LANE_ORDERED, int TT, bool SELF_SEG = false>
__global__ __launch_bounds__(TT) void tile_scatter_kernel(int P, int V, int gx, int gy, int nchunk, int tile_bits,
                                                             int stage_cap, const int32_t* __restrict__ nvis,
                                                             const uint32_t* __restrict__ rects,
                                                             const int32_t* __restrict__ ids,
                                                             const float4* __restrict__ rec,
                                                             uint32_t* __restrict__ seg_off,
                                                             int32_t* __restrict__ point_list, unsigned int list_cap, int elist_cap,
                                                             SelfSeg self) {
  // list_cap: entries the point list holds.  A speculative launch (gr_raster_forward) sizes the list before the instance
  // count is known: a chunk that would end past it writes nothing (the host then repeats the render with a larger list)
  extern __shared__ unsigned int s_cur[];  // [waves][tiles] counts -> cursors, then [stage_cap] staged chunk-local indices
  constexpr int NW = TT / WAVE, CW = BIN_CHUNK / NW;
  int v, c;
  if (!bin_block(V, nchunk, v, c)) return;
  const int tiles = gx * gy;
  uint32_t* seg = seg_off + ((int64_t)v * nchunk + c) * (tiles + 1);
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  // the wave's rectangles and ids, requested up front and before anything else is waited for (on clamped positions; what
  // lies past the view's visible count is dropped once that count has arrived): both walks below run out of registers, and
  // the trip to memory overlaps the ones of the prologue instead of following them
  const int64_t vbase = (int64_t)v * P;
  const int w_begin = c * BIN_CHUNK + wv * (BIN_CHUNK / (TT / WAVE));
  uint32_t rr[BIN_CHUNK / TT];
  int32_t ii[BIN_CHUNK / TT];
#pragma unroll
  for (int j = 0; j < BIN_CHUNK / TT; ++j) {
    const int64_t o = vbase + min(w_begin + j * WAVE + lane, P - 1);
    rr[j] = rects[o];
    ii[j] = ids[o];
  }
  const int nv = nvis[v];
  __shared__ int s_red[2][TT / WAVE];
  __shared__ unsigned int s_carry;
  unsigned int chunk_begin;
  int total;
  if (SELF_SEG) {
    const int me = v * nchunk + c;  // everything in front belongs to earlier views / chunks
    int part = 0;
    for (int i = threadIdx.x; i < me; i += TT) part += self.chunk_total[i];
    part = wave_sum_i32_dpp(part);
    if (lane == 0) s_red[0][wv] = part;
    __syncthreads();
    int front = 0;
#pragma unroll
    for (int w2 = 0; w2 < TT / WAVE; ++w2) front += s_red[0][w2];
    chunk_begin = (unsigned int)front;
    total = self.chunk_total[me];
    if (c == 0) {  // block-uniform: the view's figures for the host
      __syncthreads();
      int sum = 0, mx = 0;
      for (int i = threadIdx.x; i < nchunk; i += TT) {
        const int t = self.chunk_total[v * nchunk + i];
        sum += t;
        mx = max(mx, t);
      }
      sum = wave_sum_i32_dpp(sum);
      mx = wave_max_i32_dpp(mx);
      if (lane == 0) s_red[0][wv] = sum, s_red[1][wv] = mx;
      __syncthreads();
      if (threadIdx.x == 0) {
        int tv = 0, mv = 0;
        for (int w2 = 0; w2 < TT / WAVE; ++w2) tv += s_red[0][w2], mv = max(mv, s_red[1][w2]);
        self.totals[v] = tv;
        self.totals[V + v] = mv;
        if (self.mail != nullptr) {
          __hip_atomic_store(&self.mail[v], tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&self.mail[V + v], mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (v == 0) __hip_atomic_store(&self.mail[2 * V], self.totals[2 * V], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&self.mail[2 * V + 1 + v], self.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (total == 0) {  // an empty row for the blend
      for (int T = threadIdx.x; T <= tiles; T += TT) seg[T] = chunk_begin;
      return;
    }
    // (a chunk that would end past the list still counts and scans: the render that repeats this frame with a larger list
    // finds every row of seg_off in place; it stops before it writes any entry)
  } else {
    chunk_begin = seg[0];
    total = (int)(seg[tiles] - chunk_begin);
    if (total == 0 || seg[tiles] > list_cap) return;  // block-uniform
  }
  const bool staged = total <= stage_cap;
  unsigned int* my = s_cur + wv * tiles;
  // staged entries are 16-bit positions inside the chunk (BIN_CHUNK <= 65536); the ids are looked up on the way out
  unsigned short* stage = reinterpret_cast<unsigned short*>(s_cur + NW * tiles);
  // (elist_cap > 0) per wave: the (owner lane, tile) words of one 64-Gaussian step, behind the staging block
  unsigned int* elist = s_cur + NW * tiles + (stage_cap + 1) / 2 + wv * elist_cap;
  for (int T = threadIdx.x; T < NW * tiles; T += TT) s_cur[T] = 0u;
  __syncthreads();
  const int w_end = min(nv, w_begin + CW);
#pragma unroll
  for (int j = 0; j < CW / WAVE; ++j) {
    const int t = w_begin + j * WAVE + lane;
    rr[j] = t < w_end ? rr[j] : 0u;  // (past the view's visible count the arrays hold leftovers)
    ii[j] = t < w_end ? ii[j] : 0;
  }
  // ---- phase A: this wave's tile counts
#pragma unroll
  for (int j = 0; j < CW / WAVE; ++j) {
    const uint32_t r = rr[j];
    int x0, y0, w, h;
    if (r != 0u && rect_decode(r, ii[j], vbase, rec, gx, gy, x0, y0, w, h))
      for (int y = y0; y < y0 + h; ++y)
        for (int x = x0; x < x0 + w; ++x) atomicAdd(&my[y * gx + x], 1u);
  }
  __syncthreads();
  // ---- phase B: counts -> cursors (chunk-local when staged, global otherwise)
  if (SELF_SEG) {
    // the segment starts are this workgroup's own exclusive scan over the tiles of the counts it just made
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int T0 = 0; T0 < tiles; T0 += TT) {
      const int T = T0 + (int)threadIdx.x;
      unsigned int cw[NW];
      int n_all = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) {
        cw[w2] = T < tiles ? s_cur[w2 * tiles + T] : 0u;
        n_all += (int)cw[w2];
      }
      const int incl = wave_incl_scan_add_dpp(n_all);
      if (lane == WAVE - 1) s_red[0][wv] = incl;
      __syncthreads();
      unsigned int before = s_carry + (unsigned int)(incl - n_all);
      for (int w2 = 0; w2 < wv; ++w2) before += (unsigned int)s_red[0][w2];
      if (T < tiles) {
        seg[T] = chunk_begin + before;
        unsigned int run = before + (staged ? 0u : chunk_begin);
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
          s_cur[w2 * tiles + T] = run;
          run += cw[w2];
        }
      }
      __syncthreads();
      if (threadIdx.x == TT - 1) s_carry = before + (unsigned int)n_all;
      __syncthreads();
    }
    if (threadIdx.x == 0) seg[tiles] = chunk_begin + (unsigned int)total;
    if (chunk_begin + (unsigned int)total > list_cap) return;  // block-uniform
  } else {
    for (int T = threadIdx.x; T < tiles; T += TT) {
      unsigned int run = seg[T] - (staged ? chunk_begin : 0u);
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) {
        const unsigned int n = s_cur[w2 * tiles + T];
        s_cur[w2 * tiles + T] = run;
        run += n;
      }
    }
  }
  __syncthreads();
  // ---- phase C
  // two typed stores under a block-uniform branch (a generic pointer would turn the LDS case into flat_store)
  auto put = [&](unsigned int pos, int local, int val) {
    if (staged) stage[pos] = (unsigned short)local;
    else point_list[pos] = val;
  };
  const unsigned long long lt = (1ull << lane) - 1ull;
  // (rolled: the body is large; the preloaded values move down one register per step instead of being indexed)
#pragma unroll 1
  for (int t0 = w_begin; t0 < w_end; t0 += WAVE) {
    const int t = t0 + lane;
    int x0 = 0, y0 = 0, w = 0, h = 0;
    const int id = ii[0];
    const uint32_t r_now = rr[0];
#pragma unroll
    for (int k = 0; k + 1 < CW / WAVE; ++k) {
      rr[k] = rr[k + 1];
      ii[k] = ii[k + 1];
    }
    if (r_now != 0u && !rect_decode(r_now, id, vbase, rec, gx, gy, x0, y0, w, h)) w = h = 0;
    unsigned long long live = __ballot(w * h != 0);
    if (live == 0ull) continue;
    const int maxd = wave_max_i32_dpp(max(w, h));
    if (maxd <= 8) {
      // all requests of a row of residue classes are issued before the first returned slot is used: M LDS atomics in
      // flight instead of one round trip per class (a wave's LDS instructions still execute in issue order)
      auto walk = [&](auto mtag) {
        constexpr int M = decltype(mtag)::value;
#pragma unroll
        for (int ry = 0; ry < M; ++ry) {
          const int dy = (ry - y0) & (M - 1);
          const int rowbase = (y0 + dy) * gx + x0;
          const bool vy = dy < h;
          if (LANE_ORDERED) {
            unsigned int pos[M];
            bool act[M];
#pragma unroll
            for (int rx = 0; rx < M; ++rx) {
              const int dx = (rx - x0) & (M - 1);
              act[rx] = vy && dx < w;
              pos[rx] = 0u;
              if (act[rx]) pos[rx] = atomicAdd(&my[rowbase + dx], 1u);
            }
#pragma unroll
            for (int rx = 0; rx < M; ++rx)
              if (act[rx]) put(pos[rx], t - c * BIN_CHUNK, id);
          } else {
            for (int rx = 0; rx < M; ++rx) {
              const int dx = (rx - x0) & (M - 1);
              const bool act = vy && dx < w;
              const unsigned int tile = (unsigned int)(rowbase + dx);
              unsigned long long peers = __ballot(act);
              if (peers == 0ull) continue;
              for (int bit = 0; bit < tile_bits; ++bit) {
                const bool one = (tile >> bit) & 1u;
                const unsigned long long bal = __ballot(one);
                peers &= one ? bal : ~bal;
              }
              if (act) {
                const unsigned int base = my[tile];
                put(base + (unsigned int)__popcll(peers & lt), t - c * BIN_CHUNK, id);
                if ((peers >> lane) == 1ull) my[tile] = base + (unsigned int)__popcll(peers);  // last lane of the group
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
          }
        }
      };
      // Rectangles of 5 .. 8 tiles a side (large images): the 8 x 8 residue walk issues 64 rounds with a seventh of the lanes
      // active.  Instead every lane lists its own tiles (owner lane << 16 | tile) at its offset in a per-wave LDS list, and the
      // wave then works through the list with all lanes busy: lane i of a round holds instance i -- instances are in (owner,
      // tile) order, so the lanes of one atomic still arrive at a tile's cursor in depth order.
      int n_inst = 0, inc_inst = 0, t_inst = 0x7fffffff;
      if (maxd > 4 && elist_cap > 0) {
        n_inst = w * h;
        inc_inst = wave_incl_scan_add_dpp(n_inst);
        t_inst = __builtin_amdgcn_readlane(inc_inst, WAVE - 1);
      }
      if (maxd <= 2) walk(std::integral_constant<int, 2>{});
      else if (maxd <= 4) walk(std::integral_constant<int, 4>{});
      else if (t_inst > elist_cap) walk(std::integral_constant<int, 8>{});
      else {
        {
          unsigned int* dst = elist + (inc_inst - n_inst);
          const unsigned int tag = (unsigned int)lane << 16;
          for (int dy = 0; dy < h; ++dy)
            for (int dx = 0; dx < w; ++dx) *dst++ = tag | (unsigned int)((y0 + dy) * gx + x0 + dx);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i0 = 0; i0 < t_inst; i0 += WAVE) {
          const int i = i0 + lane;
          const bool act = i < t_inst;
          const unsigned int e = act ? elist[i] : 0u;
          const int owner = (int)(e >> 16);
          const unsigned int tile = e & 0xffffu;
          const int oid = __shfl(id, owner, WAVE);
          const int olocal = t0 + owner - c * BIN_CHUNK;
          if (LANE_ORDERED) {
            if (act) put(atomicAdd(&my[tile], 1u), olocal, oid);
          } else {
            unsigned long long peers = __ballot(act);
            for (int bit = 0; bit < tile_bits; ++bit) {
              const bool one = (tile >> bit) & 1u;
              const unsigned long long bal = __ballot(one);
              peers &= one ? bal : ~bal;
            }
            if (act) {
              const unsigned int base = my[tile];
              put(base + (unsigned int)__popcll(peers & lt), olocal, oid);
              if ((peers >> lane) == 1ull) my[tile] = base + (unsigned int)__popcll(peers);  // last lane of the group
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    } else {
      const uint32_t pr = (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)w << 16) | ((uint32_t)h << 24);  // gx, gy <= 255
      while (live) {
        const int j = __ffsll((long long)live) - 1;
        live &= live - 1ull;
        const uint32_t sr = (uint32_t)__builtin_amdgcn_readlane((int)pr, j);
        const int sid = __builtin_amdgcn_readlane(id, j);
        const int sx = (int)(sr & 255u), sy = (int)((sr >> 8) & 255u), sw = (int)((sr >> 16) & 255u), sh = (int)(sr >> 24);
        for (int y = 0; y < sh; ++y)
          for (int xb = 0; xb < sw; xb += WAVE)
            if (xb + lane < sw) {
              unsigned int* slot = &my[(sy + y) * gx + sx + xb + lane];
              if (LANE_ORDERED) {
                put(atomicAdd(slot, 1u), t0 + j - c * BIN_CHUNK, sid);
              } else {
                const unsigned int pos = *slot;  // distinct tiles: no two lanes share a slot here
                *slot = pos + 1u;
                put(pos, t0 + j - c * BIN_CHUNK, sid);
              }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  if (!staged) return;
  __syncthreads();
  // ---- phase D: the chunk's block leaves as full coalesced lines
  const int32_t* cid = ids + vbase + (int64_t)c * BIN_CHUNK;
  // (four independent LDS -> gather -> store chains in flight per thread: a rolled loop runs them one after the other)
  int i = threadIdx.x;
  for (; i + 3 * TT < total; i += 4 * TT) {
    const int a0 = cid[stage[i]], a1 = cid[stage[i + TT]], a2 = cid[stage[i + 2 * TT]], a3 = cid[stage[i + 3 * TT]];
    point_list[chunk_begin + i] = a0;
    point_list[chunk_begin + i + TT] = a1;
    point_list[chunk_begin + i + 2 * TT] = a2;
    point_list[chunk_begin + i + 3 * TT] = a3;
  }
  for (; i < total; i += TT) point_list[chunk_begin + i] = cid[stage[i]];
}

// ------------------------------------------------------------------------------------ blend
// Per-tile front-to-back blend.  A 16x16 tile is 16 cells of 4x4 pixels; 16 consecutive lanes own
// one cell, a wave owns a row of four cells.  Per batch of 256 depth-ordered Gaussians:
//   load   thread k gathers Gaussian k's 48-byte record into LDS, computes its exact cutoffs
//          (pc: power below which alpha < 1/255; rc2: squared radius beyond which power < pc) and a
//          16-bit mask of the cells its cutoff circle touches;
//   lists  wave ballots turn the masks into 16 ORDER-PRESERVING index lists (one per cell);
//   blend  every 16-lane group walks ITS OWN list -- lanes of one wave work on different
//          Gaussians in the same instruction (per-lane LDS gathers), so a wave's trip count is
//          the longest of its four cell lists (~1/4 of the batch) instead of the whole batch.
// The cutoffs only skip (pixel, Gaussian) pairs that the reference test `alpha < 1/255` skips,
// and per-pixel order is untouched, so the image stays bit-identical to the oracle.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int CELL = 4;
constexpr int NCELL = (TILE / CELL) * (TILE / CELL);  // 16

__device__ __forceinline__ float lds_f32(const float* plane, unsigned int byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(plane) + byte_off);
}
// fmaxf without the canonicalising v_max(x, x) in front (x is the result of an fma here: never a signalling NaN)
__device__ __forceinline__ float max_f32_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float min_f32_raw(float a, float b) {  // fminf, same remark (a NaN operand yields the other one)
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// FAST_EXP: alpha = opacity * 2^(power * log2 e) on the hardware exponential (v_exp_f32, 1 ulp) instead of the oracle's
// deterministic polynomial exp_det() -- 3 issue slots per pair of entries instead of 13.  The image is no longer bit-equal
// to oracle/rasterizer_oracle.c but stays within 1e-5 relative of it (tests/test_gpu_rasterizer_fast.py); opt-in
// (GR_RASTER_FAST_EXP flag of gr_raster_render_ex).
template <bool FAST_EXP>
__global__ __launch_bounds__(BLOCK) void blend_kernel(
    int P, int W, int H, int nchunk, const DevView* __restrict__ views, const uint32_t* __restrict__ seg_off,
    const int32_t* __restrict__ point_list, const float4* __restrict__ rec, float* __restrict__ out_color,
    unsigned int list_cap) {
  // One plane per field: the blend reads field f of two different entries into the two halves of a register pair
  // (ds_read_b32 x 2), which is the operand layout of the packed fp32 instructions -- no register shuffling.
  // (Plane stride 257 dwords: neither <= 255 nor a multiple of 64, so the compiler cannot fuse two fields of ONE entry
  // into a ds_read2[st64]_b32 -- that would hand back exactly the wrong pairing.)
  constexpr int PL = BLOCK + 1;
  __shared__ float s_pl[10 * PL];
  float* const s_px = s_pl, * const s_py = s_pl + PL, * const s_pc = s_pl + 2 * PL;          // centre, power cutoff
  float* const s_cx = s_pl + 3 * PL, * const s_cy = s_pl + 4 * PL, * const s_cz = s_pl + 5 * PL;  // conic
  float* const s_op = s_pl + 6 * PL;                                                          // opacity
  float* const s_cr = s_pl + 7 * PL, * const s_cg = s_pl + 8 * PL, * const s_cb = s_pl + 9 * PL;  // colour
  __shared__ unsigned short s_list[NCELL][BLOCK + 2];                   // byte offsets (4 * entry) into the planes
  __shared__ int s_cnt[NCELL][BLOCK / WAVE];      // per (cell, loading wave) counts
  __shared__ int s_alldone[BLOCK / WAVE];
  __shared__ int s_wpre[WAVE];                    // window of 64 chunks: inclusive prefix of this tile's segment lengths
  __shared__ unsigned int s_woff[WAVE];           //                      and where each segment starts in point_list
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  // speculative launch: the list was sized before the instance count was known; if it is too short nothing is drawn (the
  // host repeats the render).  The grand total is the end of the last segment of the last view.
  if (list_cap != 0xffffffffu && nchunk > 0 &&
      seg_off[(int64_t)gridDim.z * nchunk * (gx * gy + 1) - 1] > list_cap) return;
  const int v = blockIdx.z;
  const int tile = blockIdx.y * gx + blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1), lw = tid / WAVE;
  const int cell = tid / (CELL * CELL), pin = tid % (CELL * CELL);
  const int lx = (cell % (TILE / CELL)) * CELL + pin % CELL;
  const int ly = (cell / (TILE / CELL)) * CELL + pin / CELL;
  const int pxi = blockIdx.x * TILE + lx, pyi = blockIdx.y * TILE + ly;
  const bool inside = pxi < W && pyi < H;
  const float pfx = (float)pxi, pfy = (float)pyi;
  const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
  const int tiles = gx * gy;
  const int64_t goff = (int64_t)v * P;
  bool done = !inside;
  // entry BLOCK of every plane: the pad entry of odd-length lists.  Cutoff +inf: `power < pc` holds, it never blends.
  if (tid < 10) s_pl[tid * PL + BLOCK] = tid == 2 ? INFINITY : 0.0f;
  float T = 1.0f, C0 = 0.f;
  f32x2 C12 = {0.f, 0.f};
  // The tile's list = its segments of chunk 0, 1, 2, ... (depth order).  64 chunks are looked up at a time (one wave:
  // lane = chunk, two 4-byte loads give segment start and end); the batches of 256 entries are cut out of that window.
  int c_next = 0, w_pos = 0, w_total = 0;  // block-uniform
  const uint32_t* seg_col = seg_off + (int64_t)v * nchunk * (tiles + 1) + tile;
  while (true) {
    // all 256 pixels saturated?  One ballot per wave, one flag per wave, one barrier (the later barriers of the batch
    // separate this read from the next write)
    const unsigned long long live = __ballot(!done);
    if (lane == 0) s_alldone[lw] = live == 0ull ? 1 : 0;
    __syncthreads();
    if (s_alldone[0] & s_alldone[1] & s_alldone[2] & s_alldone[3]) break;
    bool exhausted = false;
    while (w_pos >= w_total) {
      if (c_next >= nchunk) {
        exhausted = true;
        break;
      }
      if (tid < WAVE) {
        const int c = c_next + tid;
        unsigned int a = 0u, b = 0u;
        if (c < nchunk) {
          a = seg_col[(int64_t)c * (tiles + 1)];
          b = seg_col[(int64_t)c * (tiles + 1) + 1];
        }
        s_woff[tid] = a;
        s_wpre[tid] = wave_incl_scan_add_dpp((int)(b - a));
      }
      __syncthreads();
      w_total = s_wpre[WAVE - 1];
      w_pos = 0;
      c_next += WAVE;
      __syncthreads();
    }
    if (exhausted) break;
    // ---- load + cutoffs + cell mask
    const int e = w_pos + tid;
    w_pos += BLOCK;
    // reach[c]: the lanes of this wave whose entry can reach cell c -- 16 lane masks in scalar registers.  The rank of an
    // entry in a cell's list is a masked bit count of that mask and the list write runs under it as the exec mask
    // (inverse ballot): no per-lane bit field is built or taken apart.
    float ctr_x = 0.0f, ctr_y = 0.0f, rc2 = -1.0f, hx2 = 0.0f, hy2 = 0.0f;  // past the end of the list: reaches no cell
    if (e < w_total) {
      int lo = 0;  // first chunk of the window whose inclusive prefix exceeds e
#pragma unroll
      for (int st = WAVE / 2; st > 0; st >>= 1)
        if (s_wpre[lo + st - 1] <= e) lo += st;
      const int before = lo ? s_wpre[lo - 1] : 0;
      const float4* r = rec + 4 * (goff + point_list[s_woff[lo] + (unsigned int)(e - before)]);
      const float4 r0 = r[0];
      const float4 co = r[1];
      const float4 col = r[2];
      // Exact skip rules (they only ever skip what the reference test `alpha < 1/255` skips):
      //   power < pc = -ln(255 op) - 1e-3   ==>   op * exp(power) < 1/255
      //   |d|^2 > |pc| * col.w              ==>   power < pc            (see preprocess)
      // NaN / non-positive opacity make every comparison false: nothing is skipped early.
      const float pc = -__logf(255.0f * co.w) - 1.0e-3f;
      rc2 = -pc * col.w;
      // exact level set of the quadratic form, per axis (see preprocess): c' = |pc| + 2e-6 rc2 covers the fp32 error
      const float cpr = -pc + 1.0e-3f + 2.0e-6f * rc2;
      hx2 = cpr * r0.z;
      hy2 = cpr * r0.w;
      ctr_x = r0.x;
      ctr_y = r0.y;
      s_px[tid] = r0.x;
      s_py[tid] = r0.y;
      s_pc[tid] = pc;
      s_cx[tid] = co.x;
      s_cy[tid] = co.y;
      s_cz[tid] = co.z;
      s_op[tid] = co.w;
      s_cr[tid] = col.x;
      s_cg[tid] = col.y;
      s_cb[tid] = col.z;
    }
    float ex2[TILE / CELL], ey2[TILE / CELL];
    bool xin[TILE / CELL], yin[TILE / CELL];
#pragma unroll
    for (int c = 0; c < TILE / CELL; ++c) {
      const float xlo = tx0 + (float)(c * CELL), ylo = ty0 + (float)(c * CELL);
      const float ex = fmaxf(fmaxf(xlo - ctr_x, ctr_x - (xlo + (float)(CELL - 1))), 0.0f);
      const float ey = fmaxf(fmaxf(ylo - ctr_y, ctr_y - (ylo + (float)(CELL - 1))), 0.0f);
      ex2[c] = ex * ex;
      ey2[c] = ey * ey;
      xin[c] = !(ex2[c] > hx2);
      yin[c] = !(ey2[c] > hy2);
    }
    // ---- order-preserving per-cell lists
    unsigned long long reach[NCELL];
    int cnt_lane = 0;  // lane c: how many entries of this wave reach cell c
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      reach[c] = __ballot(!(ex2[c % (TILE / CELL)] + ey2[c / (TILE / CELL)] > rc2) && xin[c % (TILE / CELL)] &&
                          yin[c / (TILE / CELL)]);
      asm("v_writelane_b32 %0, %1, %2" : "+v"(cnt_lane) : "s"((int)__popcll(reach[c])), "n"(c));
    }
    if (lane < NCELL) s_cnt[lane][lw] = cnt_lane;
    __syncthreads();
    // every wave adds up the four per-wave counts itself (lane c: cell c) -- no second barrier for a 16-thread scan
    int base_lane = 0, tot_lane = 0;  // lane c: where this wave's entries start in list c; the list's length
    if (lane < NCELL) {
#pragma unroll
      for (int w = 0; w < BLOCK / WAVE; ++w) {
        const int n = s_cnt[lane][w];
        base_lane += w < lw ? n : 0;
        tot_lane += n;
      }
      // lists are walked in pairs: pad an odd one with the never-hit entry
      if (lw == 0 && (tot_lane & 1)) s_list[lane][tot_lane] = (unsigned short)(4 * BLOCK);
    }
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      const int base = __builtin_amdgcn_readlane(base_lane, c);
      const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(reach[c] >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned int)reach[c], 0u));
      if (__builtin_amdgcn_inverse_ballot_w64(reach[c])) s_list[c][base + rank] = (unsigned short)(4 * tid);
    }
    __syncthreads();
    // ---- blend: each 16-lane group walks its own list
    const int len_cell = __shfl(tot_lane, (BLOCK / WAVE) * lw + lane / (CELL * CELL));  // cell = 4 lw + lane / 16
    const int n_cell = done ? 0 : len_cell;
    // Two list entries per step: everything up to alpha is evaluated for both at once with packed fp32 math
    // (v_pk_fma/mul/add_f32 -- the same IEEE operations as the scalar sequence of the oracle, two per lane-slot);
    // only the order-dependent tail (transmittance test, colour accumulation) runs entry by entry.
    const unsigned short* lp = &s_list[cell][0];
    for (int i = 0; i < n_cell && !done; i += 2) {
      const unsigned int o0 = lp[i], o1 = lp[i + 1];  // byte offsets of two list entries (an odd list ends with the pad entry)
      const f32x2 dx = f32x2{lds_f32(s_px, o0), lds_f32(s_px, o1)} - pfx, dy = f32x2{lds_f32(s_py, o0), lds_f32(s_py, o1)} - pfy;
      const f32x2 cx = {lds_f32(s_cx, o0), lds_f32(s_cx, o1)}, cy = {lds_f32(s_cy, o0), lds_f32(s_cy, o1)};
      const f32x2 cz = {lds_f32(s_cz, o0), lds_f32(s_cz, o1)}, cw = {lds_f32(s_op, o0), lds_f32(s_op, o1)};
      const float pc0 = lds_f32(s_pc, o0), pc1 = lds_f32(s_pc, o1);
      const f32x2 q = __builtin_elementwise_fma(cx * dx, dx, (cz * dy) * dy);
      const f32x2 power = __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, q, -((cy * dx) * dy));
      f32x2 al;
      if (FAST_EXP) {
        const f32x2 t = power * 1.44269504088896341f;
        al = cw * f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
      } else {
        // exp_det(), two at a time: 13 issue slots per pair (the Cephes form of rounds 1 - 5 took 20)
        const f32x2 x = {max_f32_raw(power.x, -86.0f), max_f32_raw(power.y, -86.0f)};
        const f32x2 l2e = {1.44269504088896341f, 1.44269504088896341f}, magic = {12582912.0f, 12582912.0f};
        const f32x2 t = x * l2e;
        const f32x2 tm = t + magic;
        const f32x2 nf = tm - magic;
        const f32x2 f = __builtin_elementwise_fma(x, l2e, -nf);
        f32x2 pl = {1.3264815788716078e-3f, 1.3264815788716078e-3f};
        pl = __builtin_elementwise_fma(pl, f, f32x2{9.671512059867382e-3f, 9.671512059867382e-3f});
        pl = __builtin_elementwise_fma(pl, f, f32x2{5.550733581185341e-2f, 5.550733581185341e-2f});
        pl = __builtin_elementwise_fma(pl, f, f32x2{2.4022242426872253e-1f, 2.4022242426872253e-1f});
        pl = __builtin_elementwise_fma(pl, f, f32x2{6.931470036506653e-1f, 6.931470036506653e-1f});
        pl = __builtin_elementwise_fma(pl, f, f32x2{1.0f, 1.0f});
        al = cw * f32x2{__uint_as_float(__float_as_uint(pl.x) + (__float_as_uint(tm.x) << 23)),
                        __uint_as_float(__float_as_uint(pl.y) + (__float_as_uint(tm.y) << 23))};
      }
      const float alpha0 = min_f32_raw(al.x, 0.99f), alpha1 = min_f32_raw(al.y, 0.99f);
      const bool ok0 = !(power.x > 0.0f) && !(power.x < pc0) && !(alpha0 < 1.0f / 255.0f);
      const bool ok1 = !(power.y > 0.0f) && !(power.y < pc1) && !(alpha1 < 1.0f / 255.0f);
      // A pixel that saturates leaves the walk through the loop condition, not through a `break`: the wave's control flow
      // stays one counted loop with two predicated regions.
      if (ok0) {
        const float test_T = T * (1.0f - alpha0);
        if (test_T < 0.0001f) {
          done = true;
        } else {
          const float w = alpha0 * T;
          C0 = fmaf(lds_f32(s_cr, o0), w, C0);
          C12 = __builtin_elementwise_fma(f32x2{lds_f32(s_cg, o0), lds_f32(s_cb, o0)}, f32x2{w, w}, C12);
          T = test_T;
        }
      }
      if (ok1 && !done) {
        const float test_T = T * (1.0f - alpha1);
        if (test_T < 0.0001f) {
          done = true;
        } else {
          const float w = alpha1 * T;
          C0 = fmaf(lds_f32(s_cr, o1), w, C0);
          C12 = __builtin_elementwise_fma(f32x2{lds_f32(s_cg, o1), lds_f32(s_cb, o1)}, f32x2{w, w}, C12);
          T = test_T;
        }
      }
    }
  }
  if (inside) {
    const DevView& cam = views[v];
    float* o = out_color + (int64_t)v * 3 * H * W + (int64_t)pyi * W + pxi;
    o[0] = fmaf(T, cam.bg[0], C0);
    o[(int64_t)H * W] = fmaf(T, cam.bg[1], C12.x);
    o[2 * (int64_t)H * W] = fmaf(T, cam.bg[2], C12.y);
  }
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           float v2, float v6, float v10, float v14,
                                                           uint8_t* __restrict__ present) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float z = fmaf(v2, means3D[3 * (int64_t)i],
                       fmaf(v6, means3D[3 * (int64_t)i + 1], fmaf(v10, means3D[3 * (int64_t)i + 2], v14)));
  present[i] = z > 0.2f ? 1 : 0;
}

int check_views(const gr_raster_view* h_views, int num_views) {
  GR_REQUIRE(h_views != nullptr && num_views >= 1, "need at least one view");
  GR_REQUIRE(num_views <= MAX_VIEWS, "at most %d views per call (got %d)", MAX_VIEWS, num_views);
  for (int v = 0; v < num_views; ++v) {
    GR_REQUIRE(h_views[v].image_width == h_views[0].image_width &&
                   h_views[v].image_height == h_views[0].image_height &&
                   h_views[v].sh_degree == h_views[0].sh_degree,
               "all views of one call must share image size and sh_degree");
    GR_REQUIRE(h_views[v].image_width > 0 && h_views[v].image_height > 0, "bad image size");
    GR_REQUIRE(h_views[v].sh_degree >= 0 && h_views[v].sh_degree <= 3, "sh_degree must be 0..3");
  }
  return GR_OK;
}

// ------------------------------------------------------------------------------------ self-check of the ordered outputs
// The depth sort and the tile scatter take their stable ranks from the lane order of LDS atomics (a measured property of
// the device, common.hip).  The first frames of a process are therefore CHECKED on the device: depth order (ties: id) of
// every view, and every (chunk, tile) segment of the point list in depth order.  A failure demotes the device to explicit
// ballot ranking and the stage is redone.  GR_RASTER_VERIFY=1 checks every frame.
__global__ __launch_bounds__(256) void verify_depth_order_kernel(int P, int V, const int32_t* __restrict__ nvis,
                                                                 const uint32_t* __restrict__ dfield,
                                                                 const int32_t* __restrict__ ids, int32_t* __restrict__ rank,
                                                                 int* __restrict__ bad) {
  const int v = blockIdx.y;
  const int n = nvis[v];
  const int64_t o = (int64_t)v * P;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int a = ids[o + i];
    rank[o + a] = i;
    if (i + 1 < n) {
      const int b = ids[o + i + 1];
      const uint32_t fa = dfield[o + a], fb = dfield[o + b];
      if (fa > fb || (fa == fb && a >= b)) atomicOr(bad, 1);
    }
  }
}

__global__ __launch_bounds__(256) void verify_tile_lists_kernel(int P, int V, int nchunk, int tiles,
                                                                const uint32_t* __restrict__ seg_off,
                                                                const int32_t* __restrict__ point_list,
                                                                const int32_t* __restrict__ rank, int* __restrict__ bad) {
  const int v = blockIdx.x / nchunk;
  const uint32_t* seg = seg_off + (int64_t)blockIdx.x * (tiles + 1);
  const int32_t* rk = rank + (int64_t)v * P;
  for (int t = threadIdx.x; t < tiles; t += 256) {
    int prev = -1;
    for (uint32_t e = seg[t]; e < seg[t + 1]; ++e) {
      const int r = rk[point_list[e]];
      if (r <= prev) atomicOr(bad, 2);
      prev = r;
    }
  }
}

}  // namespace
}  // namespace gr

namespace gr {
namespace {
std::atomic<int> g_frames[64];  // frames rendered per device

// The lane-ordered ranking (one ds_add_rtn per key, equal-address lanes served in lane order) rests on a measured property of
// the LDS, not on an architectural guarantee: besides the probe (common.hip), the REAL outputs of a frame -- every view's
// depth order, every per-tile list -- are checked on the device for the first three frames of a process and device and
// again on one frame in every 256 from then on (other occupancy, clocks or partition mode later in the life of a process
// would otherwise go unnoticed); a failed check demotes the process to ballot ranking and renders the frame again.
// GR_RASTER_VERIFY=1 checks every frame.
bool verify_this_frame() {
  static const bool always = getenv("GR_RASTER_VERIFY") && getenv("GR_RASTER_VERIFY")[0] == '1';
  if (lds_atomics_lane_ordered_state() != 1) return always;  // ballot ranking needs no such check (but may be asked for)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  const int f = g_frames[dev].load();
  return always || f < 3 || (f & 255) == 0;
}

// calls left before the bucket depth sort (depth_sort.hip) is tried again after one of its buckets overflowed (a scene whose
// depths crowd into a sliver of the key range: every frame would be ordered twice); per host thread
thread_local int g_bucket_cooldown = 0;
constexpr int BUCKET_COOLDOWN_FRAMES = 256;
// One call of the waiting period.  The unit is library calls that COULD have used the bucket sort (`possible`): calls of
// many views, which never take it, do not run the period down.
bool bucket_sort_allowed(bool possible) {
  if (!possible) return false;
  if (g_bucket_cooldown == 0) return true;
  --g_bucket_cooldown;
  return false;
}
void bucket_sort_overflowed() { g_bucket_cooldown = BUCKET_COOLDOWN_FRAMES; }

void frame_done() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  g_frames[dev].fetch_add(1);
}

// runs `launch_check(d_flag)` (which enqueues the check kernels), waits, returns the flag word
template <typename F>
int device_check(hipStream_t stream, F&& launch_check, int* h_flag) {
  int* d_bad = nullptr;
  GR_HIP(hipMallocAsync(reinterpret_cast<void**>(&d_bad), sizeof(int), stream));
  GR_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), stream));
  launch_check(d_bad);
  GR_HIP(hipMemcpyAsync(h_flag, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
  GR_HIP(hipStreamSynchronize(stream));
  GR_HIP(hipFreeAsync(d_bad, stream));
  return GR_OK;
}

// One host-to-device copy: the depth-overflow flag (cleared) and the camera table that follows it in the geometry buffer
// (Geom::totals layout).  The pinned staging buffer must stay untouched until the caller has synchronised the stream / event.
// Up to four cameras travel as KERNEL ARGUMENTS: a one-wave kernel writes the cleared flag words and the DevView records.
// (An H2D copy of the same 200 bytes occupies the stream for 4.6 us in front of every one-camera frame -- copy engine
// hand-over; the kernel costs a launch boundary.)  More cameras: one copy out of pinned staging, as before.
constexpr int VIEW_PACK = 4;
struct ViewPack {
  DevView v[VIEW_PACK];
};
__global__ void upload_views_kernel(ViewPack pack, int num_views, int32_t* __restrict__ d_flag) {
  const int t = threadIdx.x;
  if (t < 4) d_flag[t] = 0;
  constexpr int WORDS = sizeof(DevView) / 4;
  float* dst = reinterpret_cast<float*>(d_flag + 4);
  const float* src = reinterpret_cast<const float*>(&pack);
  for (int i = t; i < num_views * WORDS; i += blockDim.x) dst[i] = src[i];
}

static void fill_view(DevView& d, const gr_raster_view& s) {
  memcpy(d.view, s.viewmatrix, sizeof(d.view));
  memcpy(d.proj, s.projmatrix, sizeof(d.proj));
  memcpy(d.campos, s.campos, sizeof(d.campos));
  memcpy(d.bg, s.bg, sizeof(d.bg));
  d.tanx = s.tanfovx;
  d.tany = s.tanfovy;
  d.fx = (float)s.image_width / (2.0f * s.tanfovx);
  d.fy = (float)s.image_height / (2.0f * s.tanfovy);
  d.scale_mod = s.scale_modifier;
}

int upload_views(const gr_raster_view* h_views, int num_views, int32_t* d_flag, hipStream_t stream) {
  static_assert(sizeof(DevView) % 4 == 0, "DevView is copied word by word");
  if (num_views <= VIEW_PACK) {
    ViewPack pack;
    memset(&pack, 0, sizeof(pack));
    for (int v = 0; v < num_views; ++v) fill_view(pack.v[v], h_views[v]);
    hipLaunchKernelGGL(upload_views_kernel, dim3(1), dim3(WAVE), 0, stream, pack, num_views, d_flag);
    GR_LAUNCH_CHECK();
    return GR_OK;
  }
  char* raw = static_cast<char*>(pinned_scratch(0, 16 + sizeof(DevView) * num_views));
  GR_REQUIRE(raw != nullptr, "pinned staging buffer for %d views could not be allocated", num_views);
  memset(raw, 0, 16);
  DevView* stage = reinterpret_cast<DevView*>(raw + 16);
  for (int v = 0; v < num_views; ++v) fill_view(stage[v], h_views[v]);
  GR_HIP(hipMemcpyAsync(d_flag, raw, 16 + sizeof(DevView) * num_views, hipMemcpyHostToDevice, stream));
  return GR_OK;
}

}  // namespace
}  // namespace gr

using namespace gr;

static int64_t tiles_of(int width, int height) {
  return (int64_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
}

extern "C" int gr_raster_lds_atomics_lane_ordered(void) { return lds_atomics_lane_ordered_state(); }
extern "C" int gr_raster_ballot_ranking(int on) { return lds_ballot_ranking_force(on); }

extern "C" size_t gr_raster_geom_bytes(int64_t P, int num_views, int width, int height) {
  if (P < 0 || num_views < 1 || width <= 0 || height <= 0) return 0;
  return carve_geom(nullptr, P, num_views, tiles_of(width, height)).bytes;
}

extern "C" int gr_raster_debug_geom_layout(int64_t P, int num_views, int width, int height, int64_t* h_offsets) {
  GR_REQUIRE(P >= 0 && num_views >= 1 && width > 0 && height > 0 && h_offsets != nullptr, "bad argument");
  char* const origin = reinterpret_cast<char*>(uintptr_t(1) << 32);  // never dereferenced: the carver only adds to it
  const Geom g = carve_geom(origin, P, num_views, tiles_of(width, height));
  h_offsets[0] = reinterpret_cast<char*>(g.dfield) - origin;
  h_offsets[1] = reinterpret_cast<char*>(g.order_b) - origin;
  h_offsets[2] = reinterpret_cast<char*>(g.rects) - origin;
  h_offsets[3] = reinterpret_cast<char*>(g.nvis) - origin;
  return 4;
}

extern "C" size_t gr_raster_bin_bytes(int64_t total_rendered, int width, int height, int num_views) {
  if (total_rendered < 0 || width <= 0 || height <= 0 || num_views < 1) return 0;
  return carve_bin(nullptr, total_rendered, tiles_of(width, height) * num_views).bytes;
}

// How the counts of a deferred (speculative) frame reach the host:
//   mail    the counting kernel itself stores them into host-mapped pinned memory and stamps them with `seq`; the host polls
//           the stamps (a few views per call on the self-scanning path: nothing but kernels sits in the stream)
//   event   a device-to-host copy on a side stream, followed by `ev`
struct Deferred {
  hipEvent_t ev;
  volatile int32_t* mail;  // [V] totals, [V] chunk maxima, [1] far word, [V] stamps
  int seq;                 // this frame's stamp (never 0)
  int far_seq;             // the far word equals this value iff a depth overflowed the compact keys (and its negative iff a
                           // bucket of the four-launch depth sort did not fit: the order is then not valid either)
  bool by_mail;            // out: which of the two ways this frame took
  bool bucket_sort;        // in: take the four-launch depth sort
  bool self_seg;           // out: the depth sort left the chunk totals and nothing else was launched -- the scatter of this
                           // frame scans its own segments and mails the counts (tile_scatter_kernel SELF_SEG)
};

// defer_ev != nullptr: everything is enqueued, the read-back of the counts is followed by this event instead of a stream
// synchronise, and h_num_rendered is NOT filled -- the caller waits for the event and calls preprocess_collect().
static int preprocess_impl(int64_t P, int M, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities,
                           const float* scales, const float* rotations,
                           const float* cov3D_precomp, const gr_raster_view* h_views,
                           int num_views, int32_t* radii, void* geom, size_t geom_bytes,
                           int64_t* h_num_rendered, hipStream_t stream, Deferred* defer_ev) {
  int rc = check_views(h_views, num_views);
  if (rc != GR_OK) return rc;
  GR_REQUIRE(h_num_rendered != nullptr, "h_num_rendered is null");
  for (int v = 0; v <= num_views; ++v) h_num_rendered[v] = 0;
  GR_REQUIRE(P >= 0 && P < (1ll << 31) - 1, "P out of range");
  if (P == 0) {  // nothing to project, but the render call still needs the camera table (background)
    Geom g0 = carve_geom(geom, 0, num_views, tiles_of(h_views[0].image_width, h_views[0].image_height));
    if (!geom || geom_bytes < g0.bytes) {
      set_error("raster geometry buffer too small: need %zu bytes, got %zu", g0.bytes, geom_bytes);
      return GR_ERR_WORKSPACE;
    }
    rc = upload_views(h_views, num_views, g0.totals + 2 * num_views, stream);
    if (rc != GR_OK) return rc;
    GR_HIP(hipStreamSynchronize(stream));
    return GR_OK;
  }
  GR_REQUIRE(means3D && opacities && radii, "means3D / opacities / radii must be non-null");
  GR_REQUIRE((shs != nullptr) != (colors_precomp != nullptr),
             "Please provide excatly one of either SHs or precomputed colors!");
  GR_REQUIRE(((scales != nullptr && rotations != nullptr) != (cov3D_precomp != nullptr)) &&
                 ((scales != nullptr) == (rotations != nullptr)),
             "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
  const int D = h_views[0].sh_degree;
  if (shs) GR_REQUIRE(M >= (D + 1) * (D + 1), "shs has %d coefficients, sh_degree %d needs %d", M, D, (D + 1) * (D + 1));
  const int W = h_views[0].image_width, H = h_views[0].image_height;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int tiles = gx * gy;
  GR_REQUIRE(gx <= 255 && gy <= 255, "image too large: at most 255 x 255 tiles of 16 px (got %d x %d)", gx, gy);
  const size_t bin_lds = (size_t)(BIN_T / WAVE) * tiles * sizeof(unsigned int);
  GR_REQUIRE(bin_lds <= 156 * 1024, "image too large: the per-wave tile cursors (%zu bytes) do not fit in LDS", bin_lds);
  Geom g = carve_geom(geom, P, num_views, tiles);
  if (!geom || geom_bytes < g.bytes) {
    set_error("raster geometry buffer too small: need %zu bytes, got %zu", g.bytes, geom_bytes);
    return GR_ERR_WORKSPACE;
  }
  const dim3 blk(256), grd((unsigned)((P + 255) / 256));
  const bool sh16 = shs != nullptr && M == 16 && (reinterpret_cast<uintptr_t>(shs) % 16 == 0);
  // one camera, deferred frame: the camera rides in the kernel arguments of the preprocess kernel and a far depth stores
  // this call's stamp into the flag word -- no upload, no clear in front of the frame
  // (only when the counts will leave through the mailbox, i.e. the chunk rows are short enough for the self-scanning count
  // launch: the event path below reads the far flag as "non-zero", which needs the upload's clear)
  const bool cam_in_args = sh16 && num_views == 1 && defer_ev != nullptr && defer_ev->mail != nullptr &&
                           (P + BIN_CHUNK - 1) / BIN_CHUNK <= SCAN_SINGLE_ROW;
  DevView cam1;
  memset(&cam1, 0, sizeof(cam1));
  int far_seq = 1;
  if (sh16 && num_views == 1) fill_view(cam1, h_views[0]);  // the one-view kernel variant always reads the camera from its arguments
  if (cam_in_args) {
    far_seq = defer_ev->seq;
  } else {
    rc = upload_views(h_views, num_views, g.totals + 2 * num_views, stream);  // cameras + the cleared depth-overflow flag
    if (rc != GR_OK) return rc;
  }
  if (defer_ev != nullptr) defer_ev->far_seq = far_seq;
  // a deferred frame of a few views: the four-launch depth sort (a bucket that outgrows LDS raises the far word with the
  // sign flipped; the caller then takes the plain path, which always sorts in three passes)
  // (a plain call reads the counts itself: bit 1 of the far word says "overflow" there and the ordering is redone in place)
  const bool msd_possible = depth_sort_msd_possible(P, num_views, KEY_DEPTH_BITS);
  const bool msd = msd_possible && (defer_ev != nullptr ? defer_ev->bucket_sort : bucket_sort_allowed(true));
  int2* const mm_out = msd ? g.key_mm : nullptr;
#define GR_PRE(SH, COV, S16)                                                                       \
  if (S16 && num_views == 1)                                                                       \
    hipLaunchKernelGGL((preprocess_kernel<SH, COV, S16, S16>), grd, blk, 0, stream, (int)P, D, M, num_views, \
                       g.views, means3D, shs, colors_precomp, opacities, scales, rotations,        \
                       cov3D_precomp, W, H, radii, g.rec, g.dfield, g.rect_raw, g.totals + 2 * num_views, cam1, far_seq, \
                       mm_out);                                                                    \
  else                                                                                             \
  hipLaunchKernelGGL((preprocess_kernel<SH, COV, S16, false>), grd, blk, 0, stream, (int)P, D, M, num_views, \
                     g.views, means3D, shs, colors_precomp, opacities, scales, rotations,          \
                     cov3D_precomp, W, H, radii, g.rec, g.dfield, g.rect_raw, g.totals + 2 * num_views, cam1, far_seq, \
                     mm_out)
  auto run_preprocess = [&]() {
    KernelTimer timer("raster_preprocess", stream);
    if (shs && cov3D_precomp) { if (sh16) GR_PRE(true, true, true); else GR_PRE(true, true, false); }
    else if (shs) { if (sh16) GR_PRE(true, false, true); else GR_PRE(true, false, false); }
    else if (cov3D_precomp) GR_PRE(false, true, false);
    else GR_PRE(false, false, false);
  };
  run_preprocess();
  GR_LAUNCH_CHECK();
  int32_t* tot = static_cast<int32_t*>(pinned_scratch(1, sizeof(int32_t) * (2 * num_views + 2)));  // totals, far flag, chunk maxima
  GR_REQUIRE(tot != nullptr, "pinned read-back buffer could not be allocated");
  const int nchunk = (int)((P + BIN_CHUNK - 1) / BIN_CHUNK);
  int32_t h_chunk_max = 0;
  auto sort_and_count = [&](int key_bits, bool buckets = true) -> int {
    {
      KernelTimer timer("raster_depth_sort", stream);
      // visible Gaussians of every view in depth order (ties: Gaussian id): ids -> order_b, rectangles -> rects
      const bool msd_now = msd && buckets && key_bits == KEY_DEPTH_BITS;
      // ... and with the mailbox: no count / scan launch either, the scatter does both (its SELF_SEG variant)
      const bool self_seg = msd_now && defer_ev != nullptr && defer_ev->mail != nullptr && nchunk <= SCAN_SINGLE_ROW &&
                            num_views <= 4;
      const DepthSortTotals ct{g.chunk_total, BIN_CHUNK, nchunk, g.rec, gx, gy};
      int rcs = depth_sort_views(g.dfield, g.rect_raw, g.keys_a, g.keys_b, g.order_b, g.rects, g.nvis, P, num_views, key_bits,
                                 g.ds_table, g.ds_table_bytes, stream, msd_now ? g.key_mm : nullptr,
                                 (int)((P + 255) / 256), g.totals + 2 * num_views, defer_ev != nullptr ? -far_seq : 2,
                                 self_seg ? &ct : nullptr);
      if (rcs != GR_OK) return rcs;
      if (self_seg) {
        defer_ev->by_mail = true;
        defer_ev->self_seg = true;
        return GR_OK;
      }
    }
    {
      KernelTimer timer("raster_bin", stream);
      if (tiles * sizeof(unsigned int) > 64 * 1024)
        GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_count_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      hipLaunchKernelGGL(tile_count_kernel, dim3((unsigned)bin_grid(num_views, nchunk)), dim3(BIN_T),
                         tiles * sizeof(unsigned int), stream, (int)P, num_views, gx, gy, nchunk, g.nvis, g.rects, g.order_b,
                         g.rec, g.chunk_cnt, g.chunk_total);
      // chunk bases inside each view (+ the per-view totals R_v and the largest chunk total, which sizes the scatter's
      // staging block), then every chunk's per-tile segment starts
      const bool short_rows = nchunk <= SCAN_SINGLE_ROW;  // one launch does scan, totals and maxima
      const bool self_scan = short_rows && num_views <= 4;  // ... or none at all: seg_scan_kernel<true> sums what it needs
      if (self_scan) {
        const bool by_mail = defer_ev != nullptr && defer_ev->mail != nullptr;
        hipLaunchKernelGGL(seg_scan_kernel<true>, dim3((unsigned)(num_views * nchunk)), blk, 0, stream, num_views, tiles, nchunk,
                           g.chunk_cnt, g.chunk_total, g.totals, g.seg_off, by_mail ? const_cast<int32_t*>(defer_ev->mail) : nullptr,
                           by_mail ? defer_ev->seq : 0);
        if (by_mail) defer_ev->by_mail = true;
        GR_LAUNCH_CHECK();
      } else {
        if (!short_rows)
          hipLaunchKernelGGL(chunk_max_kernel, dim3(1), dim3(1024), 0, stream, num_views * nchunk, g.chunk_total, g.chunk_max);
        GR_LAUNCH_CHECK();
        int rcs = exclusive_scan_i32(g.chunk_total, g.chunk_total + (int64_t)num_views * nchunk, nchunk, num_views, nchunk,
                                     g.scan_ws, g.totals, stream, nullptr, short_rows ? g.totals + num_views : nullptr);
        if (rcs != GR_OK) return rcs;
        hipLaunchKernelGGL(seg_scan_kernel<false>, dim3((unsigned)(num_views * nchunk)), blk, 0, stream, num_views, tiles, nchunk,
                           g.chunk_cnt, g.chunk_total + (int64_t)num_views * nchunk, g.totals, g.seg_off, (int32_t*)nullptr, 0);
        GR_LAUNCH_CHECK();
      }
    }
    const bool short_rows = nchunk <= SCAN_SINGLE_ROW;
    // tot: [V] totals, [V] per-view chunk maxima (short rows), [1] depth-overflow flag, [1] chunk maximum (long rows)
    if (defer_ev != nullptr) {
      if (defer_ev->by_mail) return GR_OK;  // the counting kernel mails the counts itself
      // else: a device-to-host copy on a SIDE stream, so the scatter and the blend that follow on `stream` do not queue
      // behind it (it held the stream for 4.2 us + a 5.7 us hand-over gap, profiles/r03_single_view_timeline.txt)
      static thread_local hipStream_t side = nullptr;
      static thread_local hipEvent_t counted = nullptr;
      if (side == nullptr) GR_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      if (counted == nullptr) GR_HIP(hipEventCreateWithFlags(&counted, hipEventDisableTiming));
      GR_HIP(hipEventRecord(counted, stream));
      GR_HIP(hipStreamWaitEvent(side, counted, 0));
      GR_HIP(hipMemcpyAsync(tot, g.totals, sizeof(int32_t) * (2 * num_views + 1), hipMemcpyDeviceToHost, side));
      if (!short_rows)
        GR_HIP(hipMemcpyAsync(tot + 2 * num_views + 1, g.chunk_max, sizeof(int32_t), hipMemcpyDeviceToHost, side));
      GR_HIP(hipEventRecord(defer_ev->ev, side));
      return GR_OK;
    }
    GR_HIP(hipMemcpyAsync(tot, g.totals, sizeof(int32_t) * (2 * num_views + 1), hipMemcpyDeviceToHost, stream));
    if (!short_rows)
      GR_HIP(hipMemcpyAsync(tot + 2 * num_views + 1, g.chunk_max, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    h_chunk_max = short_rows ? tot[num_views] : tot[2 * num_views + 1];
    if (short_rows)
      for (int v = 1; v < num_views; ++v) h_chunk_max = std::max(h_chunk_max, tot[num_views + v]);
    return GR_OK;
  };
  rc = sort_and_count(KEY_DEPTH_BITS);
  if (rc != GR_OK) return rc;
  if (defer_ev != nullptr) return GR_OK;
  auto check_depth_order = [&]() -> int {  // first frames of the process: is every view really in (depth, id) order?
    if (!verify_this_frame()) return GR_OK;
    int h_bad = 0;
    int rcv = device_check(stream, [&](int* d_bad) {
      hipLaunchKernelGGL(verify_depth_order_kernel, dim3((unsigned)std::min<int64_t>((P + 255) / 256, 4096), num_views), blk, 0,
                         stream, (int)P, num_views, g.nvis, g.dfield, g.order_b, reinterpret_cast<int32_t*>(g.keys_a), d_bad);
    }, &h_bad);
    if (rcv != GR_OK) return rcv;
    if (h_bad != 0 && lds_atomics_lane_ordered_state() == 1) {
      lds_order_demote();  // the lane-order property did not hold under load: explicit ranking from now on
      return 1;
    }
    GR_REQUIRE(h_bad == 0, "depth sort produced an unsorted order (internal error)");
    return GR_OK;
  };
  if (tot[2 * num_views] & 2) {  // a bucket of the four-launch sort overflowed: three passes, now and for a while
    bucket_sort_overflowed();
    GR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(g.totals + 2 * num_views), tot[2 * num_views] & 1, 1, stream));
    rc = sort_and_count(KEY_DEPTH_BITS, false);
    if (rc != GR_OK) return rc;
  }
  if (tot[2 * num_views] != 0) {  // some depth >= 8192: redo the ordering with full-width keys
    hipLaunchKernelGGL(full_keys_kernel, dim3((unsigned)((P * num_views + 255) / 256)), blk, 0, stream, P * num_views,
                       g.rec, radii, g.dfield);
    GR_LAUNCH_CHECK();
    rc = sort_and_count(32);
    if (rc != GR_OK) return rc;
  }
  rc = check_depth_order();
  if (rc == 1) {
    rc = sort_and_count(tot[2 * num_views] != 0 ? 32 : KEY_DEPTH_BITS, false);
    if (rc != GR_OK) return rc;
    rc = check_depth_order();
    if (rc == 1) rc = GR_OK;
  }
  if (rc != GR_OK) return rc;
#undef GR_PRE
  for (int v = 0; v < num_views; ++v) h_num_rendered[v] = tot[v];
  h_num_rendered[num_views] = h_chunk_max;  // sizes the scatter's LDS staging block in gr_raster_render
  return GR_OK;
}

extern "C" int gr_raster_render(int64_t P, const gr_raster_view* h_views, int num_views,
                                const int64_t* h_num_rendered, const void* geom, size_t geom_bytes,
                                void* bin, size_t bin_bytes, float* out_color, void* stream_) {
  static const int env_flags = (getenv("GR_RASTER_FAST_EXP") && getenv("GR_RASTER_FAST_EXP")[0] == '1') ? GR_RASTER_FAST_EXP : 0;
  return gr_raster_render_ex(P, h_views, num_views, h_num_rendered, geom, geom_bytes, bin, bin_bytes, out_color, env_flags,
                             stream_);
}

extern "C" int gr_raster_preprocess(int64_t P, int M, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities,
                                    const float* scales, const float* rotations,
                                    const float* cov3D_precomp, const gr_raster_view* h_views,
                                    int num_views, int32_t* radii, void* geom, size_t geom_bytes,
                                    int64_t* h_num_rendered, void* stream_) {
  return preprocess_impl(P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, h_views, num_views,
                         radii, geom, geom_bytes, h_num_rendered, static_cast<hipStream_t>(stream_), nullptr);
}

// waits for the counts of a deferred preprocess_impl (mailbox stamps or the event): counts -> h_num_rendered;
// *far_depth = the depth-overflow flag
// (also set when a bucket of the four-launch depth sort overflowed -- either way the frame has to be redone on the plain
// path; *bucket_overflow tells the two apart)
static int preprocess_collect(int64_t P, int num_views, int64_t* h_num_rendered, bool* far_depth, Deferred& d, hipStream_t stream,
                              bool* bucket_overflow = nullptr) {
  const int nchunk = (int)((P + BIN_CHUNK - 1) / BIN_CHUNK);
  if (d.by_mail) {
    const volatile int32_t* m = d.mail;
    // the GPU is still drawing while the host spins here; no time limit of its own (a long or shared device is not an
    // error), but a stream that has drained or failed without the stamps is
    bool ready = false;
    for (unsigned long spin = 1; !ready; ++spin) {
      ready = true;
      for (int v = 0; v < num_views; ++v) ready = ready && __atomic_load_n(&m[2 * num_views + 1 + v], __ATOMIC_ACQUIRE) == d.seq;
      if (!ready && (spin & 0xffff) == 0 && hipStreamQuery(stream) != hipErrorNotReady) {
        ready = true;
        for (int v = 0; v < num_views; ++v) ready = ready && __atomic_load_n(&m[2 * num_views + 1 + v], __ATOMIC_ACQUIRE) == d.seq;
        GR_HIP(hipStreamSynchronize(stream));
        GR_REQUIRE(ready, "rasterizer: the counting kernel did not report its totals");
      }
    }
    int32_t cm = m[num_views];
    for (int v = 1; v < num_views; ++v) {
      const int32_t mv = m[num_views + v];
      cm = std::max(cm, mv);
    }
    for (int v = 0; v < num_views; ++v) h_num_rendered[v] = m[v];
    h_num_rendered[num_views] = cm;
    *far_depth = m[2 * num_views] == d.far_seq || m[2 * num_views] == -d.far_seq;
    if (bucket_overflow) *bucket_overflow = m[2 * num_views] == -d.far_seq;
    return GR_OK;
  }
  GR_HIP(hipEventSynchronize(d.ev));
  const int32_t* tot = static_cast<const int32_t*>(pinned_scratch(1, sizeof(int32_t) * (2 * num_views + 2)));
  GR_REQUIRE(tot != nullptr, "pinned read-back buffer missing");
  const bool short_rows = nchunk <= SCAN_SINGLE_ROW;
  int32_t cm = short_rows ? tot[num_views] : tot[2 * num_views + 1];
  if (short_rows)
    for (int v = 1; v < num_views; ++v) cm = std::max(cm, tot[num_views + v]);
  for (int v = 0; v < num_views; ++v) h_num_rendered[v] = tot[v];
  h_num_rendered[num_views] = cm;
  *far_depth = tot[2 * num_views] != 0;
  if (bucket_overflow) *bucket_overflow = tot[2 * num_views] < 0;
  return GR_OK;
}

// spec_entries < 0: the instance counts in h_num_rendered are this call's (the normal render).
// spec_entries >= 0: speculative launch for gr_raster_forward -- the counts are not known on the host yet; `bin` holds
// spec_entries list entries, h_num_rendered[num_views] is only a hint for the staging block, and the kernels themselves
// refuse to run past the list (tile_scatter_kernel / blend_kernel list_cap).
static int render_impl(int64_t P, const gr_raster_view* h_views, int num_views, const int64_t* h_num_rendered,
                       const void* geom, size_t geom_bytes, void* bin, size_t bin_bytes, float* out_color, int flags,
                       int64_t spec_entries, hipStream_t stream, const Deferred* defer = nullptr) {
  int rc = check_views(h_views, num_views);
  if (rc != GR_OK) return rc;
  GR_REQUIRE(out_color != nullptr && h_num_rendered != nullptr, "null argument");
  const bool spec = spec_entries >= 0;
  const int W = h_views[0].image_width, H = h_views[0].image_height;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int64_t vtiles = (int64_t)gx * gy * num_views;
  GR_REQUIRE(vtiles < (1ll << 31), "too many tiles");
  int64_t R = 0;
  if (spec) R = spec_entries;
  else
    for (int v = 0; v < num_views; ++v) R += h_num_rendered[v];
  GR_REQUIRE(R < (1ll << 31) - 1, "too many rendered instances (%lld)", (long long)R);
  const int tiles = gx * gy;
  Geom g = carve_geom(const_cast<void*>(geom), P, num_views, tiles);
  GR_REQUIRE(P == 0 || (geom && geom_bytes >= g.bytes), "geometry buffer missing or too small");
  Bin b = carve_bin(bin, R, vtiles);
  if (!bin || bin_bytes < b.bytes) {
    set_error("raster binning buffer too small: need %zu bytes, got %zu", b.bytes, bin_bytes);
    return GR_ERR_WORKSPACE;
  }
  const int32_t* point_list = b.point_list;
  const int nchunk = (int)((P + BIN_CHUNK - 1) / BIN_CHUNK);
  const unsigned int list_cap = spec ? (unsigned int)R : 0xffffffffu;
  if (R > 0 || (spec && P > 0)) {
    // LDS: per-wave tile cursors + a staging block that holds a whole chunk's instances (chunks that do not fit write
    // straight to global memory); sized for the largest chunk of this call, capped so that two workgroups share a CU
    // few views: 1024 threads per workgroup (see the kernel) -- as long as their sixteen cursor rows leave room for the staging block
    constexpr int WIDE_T = 1024;
    const bool wide = num_views <= 4 && (size_t)(WIDE_T / WAVE) * tiles * sizeof(unsigned int) <= 100 * 1024;
    const int scatter_threads = wide ? WIDE_T : BIN_T;
    const size_t cur_bytes = (size_t)(scatter_threads / WAVE) * tiles * sizeof(unsigned int);
    const int64_t cap_max = ((int64_t)(wide ? 156 : 78) * 1024 - (int64_t)cur_bytes) / 2;
    int64_t want = std::max<int64_t>(h_num_rendered[num_views], 0);
    if (spec) want = want > 0 ? want + 64 : cap_max;  // the hint is last frame's figure; a chunk that outgrows it writes straight
                                                      // to memory (a 25 % margin here cost a resident workgroup per CU: + 16 % scatter time)
    int stage_cap = (int)std::min<int64_t>(std::max<int64_t>(cap_max, 0), (want + 63) / 64 * 64);
    // + per wave the instance list of a step with rectangles of 5 .. 8 tiles a side (see the kernel), when it still fits
    constexpr int ELIST = 1024;
    const size_t base_lds = cur_bytes + ((size_t)stage_cap * sizeof(unsigned short) + 3) / 4 * 4;
    // (only for images whose rectangles are that large: at 640 x 480 the 16 KB would cost a resident workgroup per CU)
    const int elist_cap = tiles >= 4096 && base_lds + (size_t)(scatter_threads / WAVE) * ELIST * 4 <= 156 * 1024 ? ELIST : 0;
    const size_t lds = base_lds + (size_t)(scatter_threads / WAVE) * elist_cap * 4;
    bool ordered = false;
    rc = lds_atomics_lane_ordered(stream, &ordered);
    if (rc != GR_OK) return rc;
    const bool self_seg = spec && defer != nullptr && defer->self_seg;
    const SelfSeg self{g.chunk_total, g.totals, self_seg ? const_cast<int32_t*>(defer->mail) : nullptr, self_seg ? defer->seq : 0};
    auto kern = self_seg ? (wide ? (ordered ? tile_scatter_kernel<true, WIDE_T, true> : tile_scatter_kernel<false, WIDE_T, true>)
                                 : (ordered ? tile_scatter_kernel<true, BIN_T, true> : tile_scatter_kernel<false, BIN_T, true>))
                : wide   ? (ordered ? tile_scatter_kernel<true, WIDE_T> : tile_scatter_kernel<false, WIDE_T>)
                         : (ordered ? tile_scatter_kernel<true, BIN_T> : tile_scatter_kernel<false, BIN_T>);
    int tile_bits = 0;
    while ((1 << tile_bits) < tiles) ++tile_bits;
    if (lds > 64 * 1024)
      GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - 512));  // (the SELF_SEG variants hold a few static words as well)
    {
      KernelTimer timer("raster_bin", stream);
      hipLaunchKernelGGL(kern, dim3((unsigned)bin_grid(num_views, nchunk)), dim3(scatter_threads), lds, stream, (int)P, num_views, gx, gy,
                         nchunk, tile_bits, stage_cap, g.nvis, g.rects, g.order_b, g.rec, g.seg_off, b.point_list, list_cap, elist_cap,
                         self);
      GR_LAUNCH_CHECK();
    }
    if (!spec && verify_this_frame()) {  // first frames: every (chunk, tile) segment in depth order?
      int h_bad = 0;
      rc = device_check(stream, [&](int* d_bad) {
        // (the depth check of gr_raster_preprocess left the depth rank of every Gaussian in keys_a)
        hipLaunchKernelGGL(verify_tile_lists_kernel, dim3((unsigned)(num_views * nchunk)), dim3(256), 0, stream, (int)P, num_views,
                           nchunk, tiles, g.seg_off, b.point_list, reinterpret_cast<const int32_t*>(g.keys_a), d_bad);
      }, &h_bad);
      if (rc != GR_OK) return rc;
      if (h_bad != 0 && ordered) {
        lds_order_demote();
        auto kern_b = wide ? tile_scatter_kernel<false, WIDE_T> : tile_scatter_kernel<false, BIN_T>;
        if (lds > 64 * 1024)
          GR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_b), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 512));
        hipLaunchKernelGGL(kern_b, dim3((unsigned)bin_grid(num_views, nchunk)), dim3(scatter_threads), lds, stream, (int)P,
                           num_views, gx, gy, nchunk, tile_bits, stage_cap, g.nvis, g.rects, g.order_b, g.rec, g.seg_off,
                           b.point_list, list_cap, elist_cap, self);
        GR_LAUNCH_CHECK();
      } else {
        GR_REQUIRE(h_bad == 0, "tile binning produced an unsorted list (internal error)");
      }
    }
  }
  KernelTimer timer("raster_blend", stream);
  const bool fast = (flags & GR_RASTER_FAST_EXP) != 0;
  const int blend_chunks = (R > 0 || (spec && P > 0)) ? nchunk : 0;
  // GR_RASTER_SHARE (a caller that keeps another one-camera frame in flight next to this one): the blend alone fills all 32
  // wave slots of a CU (8 workgroups x 19 KB of LDS), and the next frame's front kernels -- the chain the host waits for --
  // queue behind it.  14 KB of unused dynamic LDS cap it at 4 workgroups per CU: the blend takes longer, out of sight behind
  // the other frame, and the front chain gets its slots (5 290 -> 5 450 views/s).
  const size_t blend_pad = (spec && (flags & GR_RASTER_SHARE)) ? 14000 : 0;
#define GR_BLEND(FE)                                                                                                  \
  hipLaunchKernelGGL((blend_kernel<FE>), dim3(gx, gy, num_views), dim3(BLOCK), blend_pad, stream, (int)P, W, H, blend_chunks, \
                     g.views, g.seg_off, point_list, g.rec, out_color, list_cap)
  if (fast) GR_BLEND(true); else GR_BLEND(false);
#undef GR_BLEND
  GR_LAUNCH_CHECK();
  frame_done();
  return GR_OK;
}

extern "C" int gr_raster_render_ex(int64_t P, const gr_raster_view* h_views, int num_views,
                                   const int64_t* h_num_rendered, const void* geom, size_t geom_bytes,
                                   void* bin, size_t bin_bytes, float* out_color, int flags, void* stream_) {
  return render_impl(P, h_views, num_views, h_num_rendered, geom, geom_bytes, bin, bin_bytes, out_color, flags, -1,
                     static_cast<hipStream_t>(stream_));
}

namespace gr {
namespace {
struct Pending {  // a gr_raster_forward(GR_RASTER_SPLIT) of this thread whose counts have not been collected yet
  bool open;
  int64_t P;
  int num_views;
  int64_t entries;
  Deferred d;
  hipStream_t stream;
};
thread_local Pending g_pending{false, 0, 0, 0, Deferred{nullptr, nullptr, 0, 0, false, false, false}, nullptr};
}  // namespace
}  // namespace gr

extern "C" int gr_raster_debug_bucket_cooldown(int set) {
  const int left = gr::g_bucket_cooldown;
  if (set >= 0) gr::g_bucket_cooldown = set;
  return left;
}

extern "C" int gr_raster_forward_finish(int64_t* h_num_rendered) {
  using namespace gr;
  GR_REQUIRE(g_pending.open, "gr_raster_forward_finish: no split gr_raster_forward is open on this thread");
  GR_REQUIRE(h_num_rendered != nullptr, "h_num_rendered is null");
  Pending p = g_pending;
  g_pending.open = false;
  bool far_depth = false, bucket_overflow = false;
  int rc = preprocess_collect(p.P, p.num_views, h_num_rendered, &far_depth, p.d, p.stream, &bucket_overflow);
  if (rc != GR_OK) return rc;
  if (bucket_overflow) bucket_sort_overflowed();
  int64_t R = 0;
  for (int v = 0; v < p.num_views; ++v) R += h_num_rendered[v];
  if (far_depth) return GR_RETRY_FULL;
  return R <= p.entries ? GR_OK : GR_RETRY_BIN;
}

extern "C" int gr_raster_forward(int64_t P, int M, const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, const float* rotations,
                                 const float* cov3D_precomp, const gr_raster_view* h_views, int num_views, int32_t* radii,
                                 void* geom, size_t geom_bytes, void* bin, size_t bin_bytes, float* out_color, int flags,
                                 int64_t* h_num_rendered, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(h_num_rendered != nullptr, "h_num_rendered is null");
  GR_REQUIRE(!g_pending.open, "gr_raster_forward: the previous split call of this thread was not finished");
  const int64_t stage_hint = h_num_rendered[num_views > 0 ? num_views : 0];  // in: last frame's largest chunk (0 = unknown)
  const int64_t entries = bin && bin_bytes > 512 ? (int64_t)((bin_bytes - 512) / sizeof(int32_t)) - 64 : -1;
  // (only for a few views per call: at 32 views the host's share of a 3 ms frame is nothing, and the scatter measured 16 %
  // slower when launched this way -- 0.474 vs 0.405 ms)
  if (P > 0 && num_views <= 4 && entries > 0 && entries < (1ll << 31) - 1 && !verify_this_frame()) {
    // The host is not needed between the two halves of a frame: the counts are read back behind an event while the
    // binning scatter and the blend are launched right behind the counting kernels on a list sized by the caller
    // (last frame's count + 25 %).  The host then waits for the EVENT -- the GPU is still drawing -- and only a frame whose
    // count turns out larger than the list (the kernels refuse to run past it) is rendered again.
    static thread_local hipEvent_t ev = nullptr;
    if (ev == nullptr) GR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    static thread_local int32_t* mail = nullptr;  // host-mapped, coherent: the counting kernel writes it, the host polls it
    if (mail == nullptr) {
      static const bool no_mail = getenv("GR_NO_MAILBOX") && getenv("GR_NO_MAILBOX")[0] == '1';  // (common.hip: the same switch)
      void* mp = nullptr;
      if (!no_mail && hipHostMalloc(&mp, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        memset(mp, 0, 4096);
        mail = static_cast<int32_t*>(mp);
      } else {
        (void)hipGetLastError();
      }
    }
    static std::atomic<int> frame_seq{0};
    int seq = frame_seq.fetch_add(1) + 1;
    if (seq > 0x7ffffff0) {  // (two billion frames: start over; a stale stamp of that age cannot be in flight)
      frame_seq.store(1);
      seq = 1;
    }
    Deferred d{ev, mail, seq, 1, false, bucket_sort_allowed(depth_sort_msd_possible(P, num_views, KEY_DEPTH_BITS)), false};
    int rc = preprocess_impl(P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, h_views,
                             num_views, radii, geom, geom_bytes, h_num_rendered, stream, &d);
    if (rc != GR_OK) return rc;
    h_num_rendered[num_views] = stage_hint;
    rc = render_impl(P, h_views, num_views, h_num_rendered, geom, geom_bytes, bin, bin_bytes, out_color, flags, entries, stream,
                     &d);
    if (rc != GR_OK) return rc;
    if (flags & GR_RASTER_SPLIT) {  // the caller comes back with gr_raster_forward_finish (same thread)
      g_pending = Pending{true, P, num_views, entries, d, stream};
      return GR_PENDING;
    }
    bool far_depth = false, bucket_overflow = false;
    rc = preprocess_collect(P, num_views, h_num_rendered, &far_depth, d, stream, &bucket_overflow);
    if (rc != GR_OK) return rc;
    if (bucket_overflow) bucket_sort_overflowed();
    int64_t R = 0;
    for (int v = 0; v < num_views; ++v) R += h_num_rendered[v];
    if (!far_depth) return R <= entries ? GR_OK : GR_RETRY_BIN;
    // a depth >= 8192: the ordering has to be redone on full-width keys -- take the plain path for this frame
  }
  int rc = gr_raster_preprocess(P, M, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, h_views,
                                num_views, radii, geom, geom_bytes, h_num_rendered, stream);
  if (rc != GR_OK) return rc;
  int64_t R = 0;
  for (int v = 0; v < num_views; ++v) R += h_num_rendered[v];
  if (bin == nullptr || bin_bytes < gr_raster_bin_bytes(R, h_views[0].image_width, h_views[0].image_height, num_views))
    return GR_RETRY_BIN;  // the caller allocates gr_raster_bin_bytes(sum h_num_rendered) and calls gr_raster_render_ex
  return gr_raster_render_ex(P, h_views, num_views, h_num_rendered, geom, geom_bytes, bin, bin_bytes, out_color, flags, stream);
}

extern "C" int gr_raster_mark_visible(int64_t P, const float* means3D, const float* h_viewmatrix,
                                      uint8_t* present, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(P >= 0 && h_viewmatrix != nullptr, "bad argument");
  if (P == 0) return GR_OK;
  GR_REQUIRE(means3D && present, "null argument");
  hipLaunchKernelGGL(mark_visible_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, (int)P, means3D,
                     h_viewmatrix[2], h_viewmatrix[6], h_viewmatrix[10], h_viewmatrix[14], present);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
