// lds_dma_align.hip -- does global_load_lds_dwordx4 (gfx950 LDS-DMA, 16 B per lane) accept source
// addresses that are only 8-byte (or 4-byte) aligned?  One wavefront copies 64 pieces of 16 bytes
// from src + shift bytes into LDS and writes the LDS image back out; the host compares.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_align.hip -o lds_dma_align
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void dma_kernel(const char *src, int shift, double *out) {
  __shared__ __attribute__((aligned(16))) char img[1024];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + shift + lane * 16),
                                   (__attribute__((address_space(3))) void *)img, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const double *d = (const double *)img;
  out[2 * lane] = d[2 * lane];
  out[2 * lane + 1] = d[2 * lane + 1];
}

int main() {
  const int n = 512;
  std::vector<double> h(n);
  for (int i = 0; i < n; ++i)
    h[i] = 1000.0 + i;
  char *src;
  double *out;
  hipMalloc(&src, n * sizeof(double));
  hipMalloc(&out, 128 * sizeof(double));
  hipMemcpy(src, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
  for (int shift : {0, 8, 4, 24}) {
    hipMemset(out, 0, 128 * sizeof(double));
    hipLaunchKernelGGL(dma_kernel, dim3(1), dim3(64), 0, 0, src, shift, out);
    if (hipDeviceSynchronize() != hipSuccess) {
      std::printf("shift %d: launch failed\n", shift);
      return 1;
    }
    std::vector<double> got(128);
    hipMemcpy(got.data(), out, 128 * sizeof(double), hipMemcpyDeviceToHost);
    std::vector<double> want(128);
    std::memcpy(want.data(), (const char *)h.data() + shift, 128 * sizeof(double));
    int bad = 0;
    for (int i = 0; i < 128; ++i)
      bad += std::memcmp(&got[i], &want[i], sizeof(double)) != 0;
    std::printf("source shift %2d bytes: %s (%d of 128 doubles differ)\n", shift, bad ? "MISMATCH" : "exact", bad);
  }
  return 0;
}
