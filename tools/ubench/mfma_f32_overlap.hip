// Micro-benchmark (gfx950): does f32 vector work issue beside v_mfma_f32_16x16x4_f32 of the same or of another
// wavefront, and what do the other instruction kinds of the sweep (chain_mt16.hpp) cost: v_readlane_b32,
// v_rcp_f32, v_cndmask, v_permlane32_swap.  1, 2 and 4 wavefronts per SIMD.
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 mfma_f32_overlap.hip -o mfma_f32_overlap
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
typedef float f4 __attribute__((ext_vector_type(4)));

#define ACC16 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
#define F(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define FMA8a F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define FMA8b F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define R(i) "v_rcp_f32 %" #i ", %16\n"
#define RCP16 R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)

enum { T_MFMA = 0, T_FMA, T_MFMA_FMA, T_MFMA_FMA_DEP, T_RCP, T_READLANE, T_COUNT };

template <int TEST> __global__ void k(const float *in, float *out, unsigned long long *cyc, int reps) {
  float a[16];
  for (int i = 0; i < 16; ++i)
    a[i] = in[(threadIdx.x & 63) * 16 + i];
  float b = in[threadIdx.x & 63], c = in[(threadIdx.x + 7) & 63];
  f4 m[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  int sacc = 0;
  unsigned long long t0, t1;
  STAMP(t0);
  for (int r = 0; r < reps; ++r) {
    if (TEST == T_MFMA) { // 8 MFMAs on four independent accumulators
      for (int q = 0; q < 2; ++q)
        for (int e = 0; e < 4; ++e)
          m[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m[e], 0, 0, 0);
    }
    if (TEST == T_FMA)
      asm volatile(FMA8a FMA8b FMA8a FMA8b : ACC16 : "v"(b), "v"(c));
    if (TEST == T_MFMA_FMA) { // 4 MFMAs + 32 FMAs, independent of each other, interleaved 1 : 8
      for (int e = 0; e < 4; ++e) {
        m[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m[e], 0, 0, 0);
        if (e & 1)
          asm volatile(FMA8b : ACC16 : "v"(b), "v"(c));
        else
          asm volatile(FMA8a : ACC16 : "v"(b), "v"(c));
      }
    }
    if (TEST == T_MFMA_FMA_DEP) { // the sweep's shape: 6 MFMAs, then 32 FMAs that need their result, then MFMAs that need the FMAs
      for (int e = 0; e < 4; ++e)
        m[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m[e], 0, 0, 0);
      m[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m[0], 0, 0, 0);
      m[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m[1], 0, 0, 0);
      float d = m[0][0] + m[1][1] + m[2][2] + m[3][3];
      asm volatile(FMA8a FMA8b FMA8a FMA8b : ACC16 : "v"(b), "v"(d));
      b = a[0] * 1e-30f + b;
    }
    if (TEST == T_RCP)
      asm volatile(RCP16 : ACC16 : "v"(b), "v"(c));
    if (TEST == T_READLANE) {
      for (int q = 0; q < 16; ++q)
        sacc += __builtin_amdgcn_readlane(__float_as_int(a[q]), q);
    }
  }
  STAMP(t1);
  float s = (float)sacc;
  for (int e = 0; e < 4; ++e)
    s += m[e][0] + m[e][1] + m[e][2] + m[e][3];
  for (int i = 0; i < 16; ++i)
    s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + b;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int TEST> void run(const char *name, int instr_per_rep, const float *in, float *out, unsigned long long *cyc) {
  const int reps = 400;
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps, blocks = 256;
    for (int it = 0; it < 2; ++it)
      hipLaunchKernelGGL(k<TEST>, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, reps);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks * threads / 64);
    (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    printf("%-52s %d wave/SIMD: %8.1f cyc/rep per wave -> %7.1f cyc/rep per SIMD (%d instr/rep)\n", name, wps, med / reps,
           med / reps / wps, instr_per_rep);
    fflush(stdout);
  }
}

int main() {
  float *in, *out;
  unsigned long long *cyc;
  (void)hipMalloc(&in, 64 * 17 * 4);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipMalloc(&cyc, 256 * 16 * 8);
  std::vector<float> h(64 * 17, 1.0000001f);
  (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<T_MFMA>("8 mfma_f32_16x16x4 (4 accumulators)", 8, in, out, cyc);
  run<T_FMA>("32 v_fmac_f32", 32, in, out, cyc);
  run<T_MFMA_FMA>("4 mfma + 32 v_fmac_f32, independent, interleaved", 36, in, out, cyc);
  run<T_MFMA_FMA_DEP>("6 mfma -> 32 v_fmac_f32 -> (next rep), dependent", 38, in, out, cyc);
  run<T_RCP>("16 v_rcp_f32", 16, in, out, cyc);
  run<T_READLANE>("16 v_readlane_b32 + s_add", 16, in, out, cyc);
  return 0;
}
