// Stable device radix sort of (64-bit key, 32-bit value) pairs for the photon kd-tree build (rt_photon_build.h): the one
// place where the renderer core uses a library primitive (rocPRIM's device radix sort) instead of its own kernels — a
// plain sort, kept in a translation unit of its own so that the kernels' file does not pay for the library's templates.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "rt_sort.h"

extern "C" int rt_sort_pairs_u64(void* tmp, size_t* tmp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                                 const unsigned int* vals_in, unsigned int* vals_out, size_t n, int end_bit, void* stream) {
  hipError_t e = rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned int)end_bit,
                                           (hipStream_t)stream);
  return e == hipSuccess ? 0 : (int)e;
}
