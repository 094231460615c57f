// Device-wide stable radix sort of (u64 key, i32 value) pairs.  The sort itself is rocPRIM's
// (header-only, compiled for gfx950 into this library) -- the same role cub::DeviceRadixSort plays
// in the public 3DGS rasterizer; everything around it is this repo's own kernels.
#include <cstring>

#include "common.hpp"
// (common.hpp first: rocprim's texture iterator needs <cstring>'s memset declared)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace gr {

size_t sort_pairs_temp_bytes(int64_t n) {
  size_t bytes = 0;
  if (n <= 0) return 256;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                  (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0u, 64u,
                                  (hipStream_t)0);
  return align_up(bytes + 256, 256);
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
  size_t bytes = temp_bytes;
  GR_HIP(rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n,
                                   (unsigned)begin_bit, (unsigned)end_bit, stream));
  return GR_OK;
}

// Same sort with the value stream generated on the fly: value(i) = i mod period (the depth sort's payload is the
// Gaussian id, i.e. the position inside its view) -- nobody has to write or read a 4 B/entry iota array.
namespace {
struct ModPeriod {
  int64_t period;
  __host__ __device__ int32_t operator()(int64_t i) const { return (int32_t)(i % period); }
};
}  // namespace

int sort_pairs_u64_iota(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, int64_t period,
                        int32_t* vals_out, int64_t n, int begin_bit, int end_bit, hipStream_t stream) {
  if (n <= 0) return GR_OK;
  GR_REQUIRE(end_bit > begin_bit && period > 0, "sort_pairs_u64_iota: empty bit range or period");
  auto vals_in = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), ModPeriod{period});
  size_t need = 0;
  GR_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)begin_bit,
                                   (unsigned)end_bit, stream));
  GR_REQUIRE(need <= temp_bytes, "sort temp storage too small: need %zu, have %zu", need, temp_bytes);
  size_t bytes = temp_bytes;
  GR_HIP(rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)begin_bit,
                                   (unsigned)end_bit, stream));
  return GR_OK;
}

size_t sort_pairs_u32_temp_bytes(int64_t n) {
  size_t bytes = 0;
  if (n <= 0) return 256;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, 0u, 32u,
                                  (hipStream_t)0);
  return align_up(bytes + 256, 256);
}

int sort_pairs_u32_i32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                       const int32_t* vals_in, int32_t* vals_out, int64_t n, int begin_bit,
                       int end_bit, hipStream_t stream) {
  if (n <= 0) return GR_OK;
  if (end_bit <= begin_bit) {
    GR_HIP(hipMemcpyAsync(keys_out, keys_in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream));
    GR_HIP(hipMemcpyAsync(vals_out, vals_in, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, stream));
    return GR_OK;
  }
  size_t bytes = temp_bytes;
  GR_HIP(rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n,
                                   (unsigned)begin_bit, (unsigned)end_bit, stream));
  return GR_OK;
}

}  // namespace gr
