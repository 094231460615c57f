// PointMatching / LocalGlobalRegistration correspondence extraction, one workgroup per patch pair.
//
// Replaces the ATen chains of
//   geotransformer/modules/geotransformer/point_matching.py:32-66   compute_correspondence_matrix
//   geotransformer/modules/geotransformer/point_matching.py:96-115  forward (nonzero + gathers)
//   (== local_global_registration.py:49-83, byte-identical code)
// Per patch b: E = exp(score_mat[b]) in LDS; row-wise and column-wise top-k by k rounds of "largest
// not yet taken" (ties: lowest index, strict >); corr = (rowsel & E>thr) AND/OR (colsel & E>thr),
// AND the validity mask.  A second kernel numbers the true entries in (b, i, j) order
// (torch.nonzero order) and gathers points / indices / scores.
#include "common.hpp"

namespace gr {
namespace {

constexpr int PM_T = 256;

__global__ __launch_bounds__(PM_T) void corr_matrix_kernel(
    const float* __restrict__ score, int K1, int K2, const uint8_t* __restrict__ ref_masks,
    const uint8_t* __restrict__ src_masks, int k, int mutual, float thr, uint8_t* __restrict__ corr,
    int32_t* __restrict__ counts, int scores_are_exp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ld = K2 + 1;  // +1: a thread walking down a row-major column / along a row stays conflict-free
  float* E = reinterpret_cast<float*>(smem);
  const int ldf = K2 + 4;  // byte rows padded by one dword: threads walking different rows hit different banks
  uint8_t* flag = reinterpret_cast<uint8_t*>(smem + sizeof(float) * (size_t)K1 * ld);  // bit0 row-sel, bit1 col-sel
  int* wsum = reinterpret_cast<int*>(smem + sizeof(float) * (size_t)K1 * ld + (((size_t)K1 * ldf + 15) / 16) * 16);
  const int b = blockIdx.x;
  const float* sm = score + (int64_t)b * K1 * K2;
  for (int e = threadIdx.x; e < K1 * K2; e += PM_T) {
    // point_matching.py:96 torch.exp(score_mat); compute_correspondence_matrix itself (:32-66) receives the
    // exponentiated matrix and thresholds it as is -- no log/exp round trip for that entry
    E[(e / K2) * ld + (e % K2)] = scores_are_exp ? sm[e] : expf(sm[e]);
    flag[(e / K2) * ldf + (e % K2)] = 0;
  }
  __syncthreads();
  // rows: k rounds of arg-max over the entries not yet taken (point_matching.py:40 topk(dim=2))
  for (int i = threadIdx.x; i < K1; i += PM_T) {
    for (int round = 0; round < k; ++round) {
      float best = -INFINITY;
      int bj = -1;
      for (int j = 0; j < K2; ++j) {
        const float v = E[i * ld + j];
        if (!(flag[i * ldf + j] & 1) && (v > best || bj < 0)) {
          best = v;
          bj = j;
        }
      }
      if (bj >= 0) flag[i * ldf + bj] |= 1;
    }
  }
  __syncthreads();
  // columns (point_matching.py:48 topk(dim=1))
  for (int j = threadIdx.x; j < K2; j += PM_T) {
    for (int round = 0; round < k; ++round) {
      float best = -INFINITY;
      int bi = -1;
      for (int i = 0; i < K1; ++i) {
        const float v = E[i * ld + j];
        if (!(flag[i * ldf + j] & 2) && (v > best || bi < 0)) {
          best = v;
          bi = i;
        }
      }
      if (bi >= 0) flag[bi * ldf + j] |= 2;
    }
  }
  __syncthreads();
  int n = 0;
  for (int e = threadIdx.x; e < K1 * K2; e += PM_T) {
    const int i = e / K2, j = e % K2;
    const bool over = E[i * ld + j] > thr;  // torch.gt(score_mat, confidence_threshold)
    const uint8_t f = flag[i * ldf + j];
    const bool r = (f & 1) && over, c = (f & 2) && over;
    bool m = mutual ? (r && c) : (r || c);
    m = m && ref_masks[(int64_t)b * K1 + i] && src_masks[(int64_t)b * K2 + j];
    corr[(int64_t)b * K1 * K2 + e] = m ? 1 : 0;
    n += m ? 1 : 0;
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) n += __shfl_xor(n, d, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0) wsum[threadIdx.x / WAVE] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < PM_T / WAVE; ++w) t += wsum[w];
    counts[b] = t;
  }
}

// The same for k <= 4 (GaussReg: k = 3) in ONE scan per row and per column: the k largest entries of a line, ties to the
// lowest index, are what k rounds of "largest not yet taken" (strict >) pick.  A thread keeps them in registers as a sorted
// list and bubbles every element down it (strict >, so an equal later element never displaces an earlier one); threads
// 0..127 take the rows while threads 128..255 take the columns; the selections are k indices per line instead of a flag byte
// per entry, so the matrix alone is in LDS and two workgroups share a CU.  (The k-round kernel above spent 3 x 128 dependent
// LDS round trips per line, rows and columns one after the other, on half of its threads: 4.3 ms per 16 384 patches.)
constexpr int PM_KMAX = 4;

__device__ __forceinline__ void pm_topk_line(const float* __restrict__ base, int stride, int len, int k, float thr,
                                             int* __restrict__ picks) {
  float tv[PM_KMAX];
  int ti[PM_KMAX];
#pragma unroll
  for (int s = 0; s < PM_KMAX; ++s) tv[s] = 0.f, ti[s] = -1;
  for (int t0 = 0; t0 < len; t0 += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = base[(size_t)min(t0 + u, len - 1) * stride];  // loads first; a repeat is filtered below
    // Only entries above the confidence threshold can become correspondences, and everything ranked ahead of such an entry is
    // above the threshold too: the top-k of the entries above the threshold decides exactly what the top-k of the whole line
    // decides.  After the Sinkhorn normalisation a line holds one or two of them, so most steps end here for the whole wave.
    const bool any4 = (v[0] > thr) | (v[1] > thr) | (v[2] > thr) | (v[3] > thr);
    if (!__any(any4)) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float cv = v[u];
      int ci = (t0 + u < len && cv > thr) ? t0 + u : -1;
#pragma unroll
      for (int s = 0; s < PM_KMAX; ++s) {
        // the carried element takes slot s if the slot is empty or holds a strictly smaller value; what it displaces moves on
        const bool take = s < k && ci >= 0 && (ti[s] < 0 || cv > tv[s]);
        const float ov = tv[s];
        const int oi = ti[s];
        tv[s] = take ? cv : ov;
        ti[s] = take ? ci : oi;
        cv = take ? ov : cv;
        ci = take ? oi : ci;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < PM_KMAX; ++s) picks[s] = s < k ? ti[s] : -1;
}

__global__ __launch_bounds__(PM_T) void corr_matrix_topk_kernel(
    const float* __restrict__ score, int K1, int K2, const uint8_t* __restrict__ ref_masks,
    const uint8_t* __restrict__ src_masks, int k, int mutual, float thr, uint8_t* __restrict__ corr,
    int32_t* __restrict__ counts, int scores_are_exp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ld = K2 + 1;  // +1: a thread walking down a row-major column / along a row stays conflict-free
  float* E = reinterpret_cast<float*>(smem);
  int* rpick = reinterpret_cast<int*>(smem + (sizeof(float) * (size_t)K1 * ld + 15) / 16 * 16);  // [K1][4]
  int* cpick = rpick + (size_t)K1 * PM_KMAX;                                                     // [K2][4]
  int* wsum = cpick + (size_t)K2 * PM_KMAX;
  const int b = blockIdx.x;
  const float* sm = score + (int64_t)b * K1 * K2;
  // Masked slots (three quarters of a patch of the fine matching) can never be correspondences: their outputs are written as
  // zeros without a look at the picks.  (Their scores still take part in the top-k of the live lines, as in the reference:
  // point_matching.py applies the masks after the selection.)
  uint8_t* rmk = reinterpret_cast<uint8_t*>(wsum + PM_T / WAVE);  // [K1] then [K2]
  uint8_t* cmk = rmk + K1;
  for (int i = threadIdx.x; i < K1; i += PM_T) rmk[i] = ref_masks[(int64_t)b * K1 + i];
  for (int j = threadIdx.x; j < K2; j += PM_T) cmk[j] = src_masks[(int64_t)b * K2 + j];
  __syncthreads();
  const bool quad = (K2 & 3) == 0 && (reinterpret_cast<uintptr_t>(sm) & 15) == 0 && (reinterpret_cast<uintptr_t>(corr) & 3) == 0;
  if (quad) {
    for (int q = threadIdx.x; q < K1 * K2 / 4; q += PM_T) {
      const int i = (4 * q) / K2, j = 4 * q - i * K2;
      float4 x = reinterpret_cast<const float4*>(sm)[q];
      if (!scores_are_exp) x = make_float4(expf(x.x), expf(x.y), expf(x.z), expf(x.w));
      float* d = E + i * ld + j;
      d[0] = x.x, d[1] = x.y, d[2] = x.z, d[3] = x.w;
    }
  } else {
    for (int e = threadIdx.x; e < K1 * K2; e += PM_T) {
      const int i = e / K2, j = e % K2;
      E[i * ld + j] = scores_are_exp ? sm[e] : expf(sm[e]);
    }
  }
  __syncthreads();
  // lines: rows 0..K1-1 then columns 0..K2-1, dealt out over the workgroup (with K1 = K2 = 128: half the threads each)
  for (int l = threadIdx.x; l < K1 + K2; l += PM_T) {
    const bool live = l < K1 ? rmk[l] != 0 : cmk[l - K1] != 0;  // a masked line's picks are never looked at
    if (!live) continue;
    if (l < K1) pm_topk_line(E + (size_t)l * ld, 1, K2, k, thr, rpick + (size_t)l * PM_KMAX);
    else pm_topk_line(E + (l - K1), ld, K1, k, thr, cpick + (size_t)(l - K1) * PM_KMAX);
  }
  __syncthreads();
  int n = 0;
  auto entry = [&](int i, int j) -> bool {
    if (!(rmk[i] && cmk[j])) return false;
    const bool over = E[i * ld + j] > thr;  // torch.gt(score_mat, confidence_threshold)
    const int4 rp = *reinterpret_cast<const int4*>(rpick + (size_t)i * PM_KMAX);
    const int4 cp = *reinterpret_cast<const int4*>(cpick + (size_t)j * PM_KMAX);
    const bool r = over && (rp.x == j || rp.y == j || rp.z == j || rp.w == j);
    const bool c = over && (cp.x == i || cp.y == i || cp.z == i || cp.w == i);
    return mutual ? (r && c) : (r || c);
  };
  if (quad) {
    for (int q = threadIdx.x; q < K1 * K2 / 4; q += PM_T) {
      const int i = (4 * q) / K2, j = 4 * q - i * K2;
      unsigned word = 0u;
      if (rmk[i] && (cmk[j] | cmk[j + 1] | cmk[j + 2] | cmk[j + 3])) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool m = entry(i, j + c);
          word |= m ? 1u << (8 * c) : 0u;
          n += m ? 1 : 0;
        }
      }
      reinterpret_cast<unsigned*>(corr + (int64_t)b * K1 * K2)[q] = word;
    }
  } else {
    for (int e = threadIdx.x; e < K1 * K2; e += PM_T) {
      const bool m = entry(e / K2, e % K2);
      corr[(int64_t)b * K1 * K2 + e] = m ? 1 : 0;
      n += m ? 1 : 0;
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) n += __shfl_xor(n, d, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0) wsum[threadIdx.x / WAVE] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < PM_T / WAVE; ++w) t += wsum[w];
    counts[b] = t;
  }
}

size_t corr_topk_lds_bytes(int K1, int K2) {
  return (sizeof(float) * (size_t)K1 * (K2 + 1) + 15) / 16 * 16 + sizeof(int) * (size_t)(K1 + K2) * PM_KMAX + 64 +
         (((size_t)K1 + K2 + 15) / 16) * 16;  // + the patch's masks
}

// emit the true entries of patch b in row-major order at offsets[b] (exclusive scan of counts)
__global__ __launch_bounds__(PM_T) void corr_gather_kernel(
    const float* __restrict__ score, int K1, int K2, const uint8_t* __restrict__ corr,
    const int32_t* __restrict__ offsets, const float* __restrict__ ref_pts, const float* __restrict__ src_pts,
    const int64_t* __restrict__ ref_idx, const int64_t* __restrict__ src_idx, const float* __restrict__ global_scores,
    int use_global, float* __restrict__ o_ref_pts, float* __restrict__ o_src_pts, int64_t* __restrict__ o_ref_idx,
    int64_t* __restrict__ o_src_idx, float* __restrict__ o_scores) {
  __shared__ int wsum[PM_T / WAVE];
  const int b = blockIdx.x;
  const int total = K1 * K2;
  const int per = (total + PM_T - 1) / PM_T;  // contiguous chunk per thread keeps row-major order
  const int e0 = threadIdx.x * per, e1 = min(total, e0 + per);
  const uint8_t* cm = corr + (int64_t)b * total;
  // a thread's chunk as 16-byte vectors when it is 64 bytes (128 x 128 patches, 256 threads): the 0 / 1 bytes are counted
  // with popcounts and the set ones are walked bit by bit -- byte loads made this kernel 0.8 ms per 16 384 patches
  const bool vec = per == 64 && e1 - e0 == 64 && ((reinterpret_cast<uintptr_t>(cm) + e0) & 15) == 0;
  uint4 cv[4] = {};
  int n = 0;
  if (vec) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cv[q] = reinterpret_cast<const uint4*>(cm + e0)[q];
      n += __popc(cv[q].x) + __popc(cv[q].y) + __popc(cv[q].z) + __popc(cv[q].w);
    }
  } else {
    for (int e = e0; e < e1; ++e) n += cm[e];
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  int inc = n;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    int v = __shfl_up(inc, d, WAVE);
    if (lane >= d) inc += v;
  }
  if (lane == WAVE - 1) wsum[w] = inc;
  __syncthreads();
  int base = offsets[b];
  for (int u = 0; u < w; ++u) base += wsum[u];
  int pos = base + inc - n;
  const float gsc = use_global ? global_scores[b] : 1.0f;
  auto emit = [&](int e) {
    const int i = e / K2, j = e % K2;
    const int64_t ri = (int64_t)b * K1 + i, sj = (int64_t)b * K2 + j;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      o_ref_pts[3 * (int64_t)pos + d] = ref_pts[3 * ri + d];
      o_src_pts[3 * (int64_t)pos + d] = src_pts[3 * sj + d];
    }
    o_ref_idx[pos] = ref_idx[ri];
    o_src_idx[pos] = src_idx[sj];
    float s = expf(score[(int64_t)b * total + e]);
    if (use_global) s = s * gsc;  // point_matching.py:103
    o_scores[pos] = s;            // :105 (times corr_mat.float() == 1 here)
    ++pos;
  };
  if (vec) {
    if (n == 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned wd[4] = {cv[q].x, cv[q].y, cv[q].z, cv[q].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned bits = wd[c];
        while (bits) {
          const int k = (__ffs((int)bits) - 1) >> 3;  // bytes are 0 or 1: bit 0 of byte k
          emit(e0 + 16 * q + 4 * c + k);
          bits &= bits - 1u;
        }
      }
    }
  } else {
    for (int e = e0; e < e1; ++e)
      if (cm[e]) emit(e);
  }
}

size_t corr_lds_bytes(int K1, int K2) {
  return sizeof(float) * (size_t)K1 * (K2 + 1) + (((size_t)K1 * (K2 + 4) + 15) / 16) * 16 + 64;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" size_t gr_point_matching_workspace_bytes(int64_t batch) {
  if (batch < 0) return 0;
  return align_up((size_t)(2 * batch + 2) * sizeof(int32_t) + scan_ws_ints(batch) * sizeof(int32_t) + 1024, 256);
}

static int corr_matrix_impl(int scores_are_exp, const float* score_mat, int64_t batch, int64_t k1, int64_t k2,
                              const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int k, int mutual,
                              float confidence_threshold, uint8_t* corr_mat, int64_t* h_num_corr, void* ws,
                              size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (h_num_corr) *h_num_corr = 0;
  GR_REQUIRE(batch >= 0 && k1 > 0 && k2 > 0 && k1 <= 1024 && k2 <= 1024, "bad patch sizes");
  GR_REQUIRE(k >= 1 && k <= k1 && k <= k2, "k must be in [1, min(K1, K2)]");
  if (batch == 0) return GR_OK;
  GR_REQUIRE(score_mat && ref_knn_masks && src_knn_masks && corr_mat, "null argument");
  if (!ws || ws_bytes < gr_point_matching_workspace_bytes(batch)) {
    set_error("point_matching workspace too small");
    return GR_ERR_WORKSPACE;
  }
  const bool topk = k <= PM_KMAX;  // one scan per line (see corr_matrix_topk_kernel); larger k: k rounds per line
  const size_t lds = topk ? corr_topk_lds_bytes((int)k1, (int)k2) : corr_lds_bytes((int)k1, (int)k2);
  GR_REQUIRE(lds <= 160 * 1024, "patch %lld x %lld does not fit in LDS", (long long)k1, (long long)k2);
  if (lds > 64 * 1024)
    GR_HIP(hipFuncSetAttribute(topk ? reinterpret_cast<const void*>(&corr_matrix_topk_kernel)
                                    : reinterpret_cast<const void*>(&corr_matrix_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int32_t* counts = static_cast<int32_t*>(ws);
  int32_t* offsets = counts + batch;
  int32_t* total = offsets + batch;
  int32_t* scan_ws = total + 2;
  if (topk)
    hipLaunchKernelGGL(corr_matrix_topk_kernel, dim3((unsigned)batch), dim3(PM_T), lds, stream, score_mat, (int)k1, (int)k2,
                       ref_knn_masks, src_knn_masks, k, mutual, confidence_threshold, corr_mat, counts, scores_are_exp);
  else
    hipLaunchKernelGGL(corr_matrix_kernel, dim3((unsigned)batch), dim3(PM_T), lds, stream, score_mat, (int)k1, (int)k2,
                       ref_knn_masks, src_knn_masks, k, mutual, confidence_threshold, corr_mat, counts, scores_are_exp);
  GR_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(counts, offsets, batch, 1, batch, scan_ws, total, stream);
  if (rc != GR_OK) return rc;
  if (h_num_corr) {
    int32_t t = 0;
    GR_HIP(hipMemcpyAsync(&t, total, sizeof(t), hipMemcpyDeviceToHost, stream));
    GR_HIP(hipStreamSynchronize(stream));
    *h_num_corr = t;
  }
  return GR_OK;
}

extern "C" int gr_corr_matrix(const float* score_mat, int64_t batch, int64_t k1, int64_t k2,
                              const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int k, int mutual,
                              float confidence_threshold, uint8_t* corr_mat, int64_t* h_num_corr, void* ws,
                              size_t ws_bytes, void* stream_) {
  return corr_matrix_impl(0, score_mat, batch, k1, k2, ref_knn_masks, src_knn_masks, k, mutual, confidence_threshold,
                          corr_mat, h_num_corr, ws, ws_bytes, stream_);
}

extern "C" int gr_corr_matrix_exp(const float* exp_score_mat, int64_t batch, int64_t k1, int64_t k2,
                                  const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int k, int mutual,
                                  float confidence_threshold, uint8_t* corr_mat, int64_t* h_num_corr, void* ws,
                                  size_t ws_bytes, void* stream_) {
  return corr_matrix_impl(1, exp_score_mat, batch, k1, k2, ref_knn_masks, src_knn_masks, k, mutual, confidence_threshold,
                          corr_mat, h_num_corr, ws, ws_bytes, stream_);
}

extern "C" int gr_corr_gather(const float* score_mat, int64_t batch, int64_t k1, int64_t k2, const uint8_t* corr_mat,
                              const float* ref_knn_points, const float* src_knn_points,
                              const int64_t* ref_knn_indices, const int64_t* src_knn_indices,
                              const float* global_scores, int use_global_score, float* out_ref_points,
                              float* out_src_points, int64_t* out_ref_indices, int64_t* out_src_indices,
                              float* out_scores, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (batch == 0) return GR_OK;
  GR_REQUIRE(score_mat && corr_mat && ref_knn_points && src_knn_points && ref_knn_indices && src_knn_indices,
             "null argument");
  GR_REQUIRE(!use_global_score || global_scores, "global_scores is null");
  if (!ws || ws_bytes < gr_point_matching_workspace_bytes(batch)) {
    set_error("point_matching workspace too small");
    return GR_ERR_WORKSPACE;
  }
  const int32_t* offsets = static_cast<const int32_t*>(ws) + batch;  // written by gr_corr_matrix
  hipLaunchKernelGGL(corr_gather_kernel, dim3((unsigned)batch), dim3(PM_T), 0, stream, score_mat, (int)k1, (int)k2,
                     corr_mat, offsets, ref_knn_points, src_knn_points, ref_knn_indices, src_knn_indices, global_scores,
                     use_global_score, out_ref_points, out_src_points, out_ref_indices, out_src_indices, out_scores);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
