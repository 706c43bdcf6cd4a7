// Checks: with U, V held in the 32x32 f32 MFMA C/D layout (lane (j,h), reg p:
// element (row (p&3) + 8 (p>>2) + 4 h, col j)), sum_p mfma_32x32x2f32(U[p], V[p])
// equals U^T V, again in C/D layout.  Asymmetric integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ inline int crow(int p, int h) { return (p & 3) + 8 * (p >> 2) + 4 * h; }
__global__ void k(const float* U, const float* V, float* out) {  // column-major 32x32
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  f32x16 u, v, acc;
  for (int p = 0; p < 16; ++p) { u[p] = U[j * 32 + crow(p, h)]; v[p] = V[j * 32 + crow(p, h)]; acc[p] = 0.f; }
  for (int p = 0; p < 16; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(u[p], v[p], acc, 0, 0, 0);
  for (int p = 0; p < 16; ++p) out[j * 32 + crow(p, h)] = acc[p];
}
int main() {
  std::vector<float> U(1024), V(1024), R(1024), O(1024);
  for (int c = 0; c < 32; ++c) for (int r = 0; r < 32; ++r) { U[c*32+r] = (float)((r * 7 + c * 3) % 11 - 5); V[c*32+r] = (float)((r * 5 + c * 13) % 9 - 4); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int r = 0; r < 32; ++r) s += U[i*32+r] * V[j*32+r]; R[j*32+i] = s; }
  float *dU, *dV, *dO; hipMalloc(&dU, 4096); hipMalloc(&dV, 4096); hipMalloc(&dO, 4096);
  hipMemcpy(dU, U.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dV, V.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dU, dV, dO); hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) if (O[i] != R[i]) ++bad;
  printf("UtV via MFMA C-layout operands: %d mismatches of 1024\n", bad);
  return bad != 0;
}
