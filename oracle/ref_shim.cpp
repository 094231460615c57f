// TEST INFRASTRUCTURE ONLY -- never linked or loaded by the product path.
//
// C-ABI shim around the REFERENCE's own torch-free core, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile, target
// `_ref/libgaussreg_ref.so`).  Nothing of the reference is copied here: this
// file only declares two `extern "C"` entry points that marshal raw pointers
// into the std::vector arguments the reference functions take, exactly the
// way the reference's own ATen glue does
// (geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:29-52,
//  geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:20-38).
//
// Used by: tests/golden/gen_golden_ext.py (fixture generation, this container
// only) and tests (when the .so is present) to validate oracle/ restatements;
// bench.py may time it as cpu_baseline.kind == "reference".
#include <cstdint>
#include <cstring>
#include <vector>

#include "cpu/radius_neighbors/radius_neighbors_cpu.h"
#include "cpu/grid_subsampling/grid_subsampling_cpu.h"

extern "C" {

// Returns max_count (row width).  On the first call pass out == nullptr to get
// the width only; the result buffer is cached in a thread-local so the second
// call just copies (keeps the shim O(1) extra work).
static thread_local std::vector<long> g_last_neighbors;

int64_t ref_radius_neighbors(const float* q, int64_t nq, const float* s, int64_t ns,
                             const int64_t* q_lengths, const int64_t* s_lengths, int64_t batch,
                             float radius) {
  std::vector<PointXYZ> vq(reinterpret_cast<const PointXYZ*>(q),
                           reinterpret_cast<const PointXYZ*>(q) + nq);
  std::vector<PointXYZ> vs(reinterpret_cast<const PointXYZ*>(s),
                           reinterpret_cast<const PointXYZ*>(s) + ns);
  std::vector<long> ql(q_lengths, q_lengths + batch);
  std::vector<long> sl(s_lengths, s_lengths + batch);
  g_last_neighbors.clear();
  radius_neighbors_cpu(vq, vs, ql, sl, g_last_neighbors, radius);
  return nq > 0 ? static_cast<int64_t>(g_last_neighbors.size() / nq) : 0;
}

void ref_radius_neighbors_fetch(int64_t* out, int64_t count) {
  std::memcpy(out, g_last_neighbors.data(), sizeof(int64_t) * count);
}

// out_points must hold 3*n floats (worst case M == N).  Returns total M.
int64_t ref_grid_subsampling(const float* pts, int64_t n, const int64_t* lengths, int64_t batch,
                             float voxel, float* out_points, int64_t* out_lengths) {
  std::vector<PointXYZ> vp(reinterpret_cast<const PointXYZ*>(pts),
                           reinterpret_cast<const PointXYZ*>(pts) + n);
  std::vector<PointXYZ> vsp;
  std::vector<long> vl(lengths, lengths + batch);
  std::vector<long> vsl;
  grid_subsampling_cpu(vp, vsp, vl, vsl, voxel);
  std::memcpy(out_points, vsp.data(), sizeof(float) * 3 * vsp.size());
  for (int64_t b = 0; b < batch; ++b) out_lengths[b] = vsl[b];
  return static_cast<int64_t>(vsp.size());
}

}  // extern "C"
