// stream_fill.hpp -- zero-fill and device-to-device copy as KERNELS.
//
// Every entry point of the library only enqueues kernels on the caller's stream, so that a Newton
// loop can be captured in a hipGraph.  hipMemsetAsync / hipMemcpyAsync would become memset / memcpy
// NODES of the captured graph, and a memset node was seen (ROCm 7.2, MI355X) to leave its target
// untouched when the graph is replayed on the legacy default stream: the regularization flags of the
// Newton-KKT step stayed set and every problem came back NONPOSITIVE_REGULARIZATION.  Kernel nodes
// replay correctly on any stream.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace sipamd {

// Makes `device` the calling thread's current HIP device for the lifetime of the guard and restores
// the previous one afterwards.  Every compute entry point of the C ABI holds one: a plan launches on
// ITS device whatever the caller's current device is (a NULL stream then means that device's default
// stream), and the caller's current device is left as it was found.  hipSetDevice is thread-local
// state, not a stream operation, so it is legal during stream capture.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device)
      err = hipSetDevice(device);
    else if (err == hipSuccess)
      prev = -1; // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0)
      (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// 4-byte words (every region filled or copied here is int32 / float / double data, 4-byte aligned)
static __global__ void __launch_bounds__(256) fill_zero_kernel(uint32_t *__restrict__ dst, long words) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < words; k += (long)gridDim.x * blockDim.x)
    dst[k] = 0u;
}

static __global__ void __launch_bounds__(256)
copy_words_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, long words) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < words; k += (long)gridDim.x * blockDim.x)
    dst[k] = src[k];
}

inline unsigned fill_grid(long words) {
  const long blocks = (words + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
}

// bytes must be a multiple of 4
inline hipError_t zero_async(void *dst, size_t bytes, hipStream_t s) {
  if (bytes == 0)
    return hipSuccess;
  hipLaunchKernelGGL(fill_zero_kernel, dim3(fill_grid((long)(bytes / 4))), dim3(256), 0, s, (uint32_t *)dst,
                     (long)(bytes / 4));
  return hipGetLastError();
}

inline hipError_t copy_async(void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes == 0)
    return hipSuccess;
  hipLaunchKernelGGL(copy_words_kernel, dim3(fill_grid((long)(bytes / 4))), dim3(256), 0, s, (uint32_t *)dst,
                     (const uint32_t *)src, (long)(bytes / 4));
  return hipGetLastError();
}

} // namespace sipamd
