// copy_f4.hip -- the "achievable HBM" denominator of bench.py: a plain 16-bytes-per-lane device-to-device
// copy, the access pattern MI355X_MICROARCH.md quotes at 6.29 TB/s (read + write), as an in-tree kernel instead of
// torch.Tensor.copy_ (5.2 TB/s on this image).  Not part of the product: bench.py loads libcopy_f4.so through ctypes for its
// untimed context leg; build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libcopy_f4.so tools/ubench/copy_f4.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));

// one workgroup walks UNROLL consecutive 16 B x 256-lane slabs per trip, grid-stride: every wave instruction is a fully
// coalesced 1 KiB access, the loads of a trip are all in flight before the first store
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_f4_kernel(f4 *__restrict__ dst, const f4 *__restrict__ src, size_t n16) {
  const size_t per_trip = (size_t)256 * UNROLL;
  const size_t stride = (size_t)gridDim.x * per_trip;
  for (size_t base = (size_t)blockIdx.x * per_trip; base < n16; base += stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      if (i < n16) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      if (i < n16) {
        if (NT)
          __builtin_nontemporal_store(v[u], dst + i);
        else
          dst[i] = v[u];
      }
    }
  }
}
}  // namespace

// bytes: a multiple of 16, both pointers 16-byte aligned.  variant: 0 = plain, 1 = non-temporal loads and stores.
// blocks = 0: 256 CUs x 8.  Returns a hipError_t.
extern "C" int copy_f4(void *dst, const void *src, size_t bytes, int variant, int blocks, void *stream) {
  if ((bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return (int)hipErrorInvalidValue;
  const size_t n16 = bytes / 16;
  const int grid = blocks > 0 ? blocks : 256 * 8;
  hipStream_t s = (hipStream_t)stream;
  if (variant == 1)
    hipLaunchKernelGGL((copy_f4_kernel<4, true>), dim3(grid), dim3(256), 0, s, (f4 *)dst, (const f4 *)src, n16);
  else
    hipLaunchKernelGGL((copy_f4_kernel<4, false>), dim3(grid), dim3(256), 0, s, (f4 *)dst, (const f4 *)src, n16);
  return (int)hipGetLastError();
}
