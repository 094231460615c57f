// Stand-in position descriptors for the configs[4] harness (gaussreg_amd/pair_pipeline.py, bench.py `pairs`).
//
// NOT part of the reference's hot path and not a replacement of anything in it: the pretrained weights are not available
// offline, so the pair pipeline replaces the two LEARNED feature tensors by random Fourier features of the points'
// coordinates (pair_pipeline.py module docstring).  Stock PyTorch built them with seven elementwise launches per call
// (einsum, add, matmul, add, cos, mul, mul): 21 % of the pair path's kernel time in round 3 was ATen glue of this kind.
// One launch here: out[row][c] = mask[row] * scale * cos((T[tid[row]] p[row]) . W[:, c] + b[c])  (optionally L2-normalised).
#include "common.hpp"

namespace gr {
namespace {

// one wave per point: lanes stride the channels (coalesced stores)
__global__ __launch_bounds__(256) void standin_descriptor_kernel(const float* __restrict__ pts, int64_t n, const float* __restrict__ T,
                                                                 const int32_t* __restrict__ tid, const uint8_t* __restrict__ mask,
                                                                 const float* __restrict__ W, const float* __restrict__ b, int C,
                                                                 float scale, int normalize, float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  if (row >= n) return;
  float x = pts[3 * row], y = pts[3 * row + 1], z = pts[3 * row + 2];
  if (T != nullptr && tid != nullptr && tid[row] >= 0) {  // rigid / similarity transform of this row's pair: 3 x 4, row-major
    const float* t = T + 12 * (int64_t)tid[row];
    const float nx = (t[0] * x + t[1] * y + t[2] * z) + t[3];
    const float ny = (t[4] * x + t[5] * y + t[6] * z) + t[7];
    const float nz = (t[8] * x + t[9] * y + t[10] * z) + t[11];
    x = nx, y = ny, z = nz;
  }
  const bool keep = mask == nullptr || mask[row] != 0;
  constexpr float INV_2PI = 0.15915494309189535f;
  auto feature = [&](int c) {
    const float ph = (x * W[c] + y * W[C + c] + z * W[2 * C + c]) + b[c];
    const float r = ph * INV_2PI;
    return __builtin_amdgcn_cosf(r - floorf(r));  // v_cos_f32 takes revolutions
  };
  float f = keep ? scale : 0.0f;
  if (normalize) {  // F.normalize(., p=2, dim=1): the scale cancels
    float s2 = 0.f;
    for (int c = lane; c < C; c += WAVE) {
      const float v = feature(c);
      s2 += v * v;
    }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) s2 += __shfl_xor(s2, d, WAVE);
    f = keep ? 1.0f / fmaxf(sqrtf(s2), 1e-12f) : 0.0f;
  }
  for (int c = lane; c < C; c += WAVE) out[row * C + c] = feature(c) * f;
}

}  // namespace
}  // namespace gr

using namespace gr;

extern "C" int gr_standin_descriptors(const float* pts, int64_t n, const float* transforms, const int32_t* transform_id,
                                      const uint8_t* mask, const float* w, const float* b, int64_t c, float scale, int normalize,
                                      float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GR_REQUIRE(n >= 0 && c >= 1 && c < (1 << 20), "bad sizes");
  if (n == 0) return GR_OK;
  GR_REQUIRE(pts && w && b && out, "null argument");
  KernelTimer timer("standin", stream);
  hipLaunchKernelGGL(standin_descriptor_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, pts, n, transforms, transform_id,
                     mask, w, b, (int)c, scale, normalize, out);
  GR_LAUNCH_CHECK();
  return GR_OK;
}
