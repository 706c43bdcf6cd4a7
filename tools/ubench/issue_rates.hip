// Micro-benchmark (gfx950): issue cost of the instruction kinds the fp64 Riccati kernels are made of,
// at 1, 2 and 4 wavefronts per SIMD.  Answers, for the layout decision in DESIGN.md section 8:
//   * does a second wavefront on the SIMD raise the fp64 DPP-FMA rate above the 5 cycles / instruction
//     a lone wavefront gets (the DP pipe takes 16 lanes per clock = 4 cycles / instruction)?
//   * does a wavefront with only 32 of its 64 lanes enabled (EXEC) issue fp64 instructions faster?
//   * what do v_permlane32_swap, v_mov_b32 and the f64 MFMA cost, and does the f64 MFMA overlap fp64
//     VALU work of the same wavefront?
// hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")

#define ACC16 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
#define D(i, l) "v_fmac_f64_dpp %" #i ", %16, %17 row_newbcast:" #l " row_mask:0xf bank_mask:0xf\n"
#define DPP16 D(0, 0) D(1, 1) D(2, 2) D(3, 3) D(4, 4) D(5, 5) D(6, 6) D(7, 7) D(8, 8) D(9, 9) D(10, 10) D(11, 11) D(12, 12) D(13, 13) D(14, 14) D(15, 15)
#define DH(i, l) "v_fmac_f64_dpp %" #i ", %16, %17 row_newbcast:" #l " row_mask:0x3 bank_mask:0xf\n"
#define DPPH16 DH(0, 0) DH(1, 1) DH(2, 2) DH(3, 3) DH(4, 4) DH(5, 5) DH(6, 6) DH(7, 7) DH(8, 8) DH(9, 9) DH(10, 10) DH(11, 11) DH(12, 12) DH(13, 13) DH(14, 14) DH(15, 15)
#define F(i) "v_fmac_f64 %" #i ", %16, %17\n"
#define FMA16 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define MV(i) "v_mov_b32 %" #i ", %16\n"
#define MOV16 MV(0) MV(1) MV(2) MV(3) MV(4) MV(5) MV(6) MV(7) MV(8) MV(9) MV(10) MV(11) MV(12) MV(13) MV(14) MV(15)

enum { T_DPP = 0, T_DPP_HALF_EXEC, T_DPP_ROWMASK, T_FMA, T_FMA_HALF_EXEC, T_SWAP, T_MOV32, T_MFMA, T_MFMA_DPP, T_DPP_MOV, T_COUNT };

template <int TEST>
__global__ void k(const double *in, double *out, unsigned long long *cyc, int reps) {
  double a[16];
  for (int i = 0; i < 16; ++i)
    a[i] = in[(threadIdx.x & 63) * 16 + i];
  double b = in[threadIdx.x & 63], c = in[(threadIdx.x + 7) & 63];
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  int w0 = threadIdx.x, w1 = threadIdx.x * 3, m0 = 0, m1 = 0;
  int iv[16];
  for (int i = 0; i < 16; ++i)
    iv[i] = threadIdx.x + i;
  unsigned long long t0, t1;
  // half-enabled wavefront: a real branch, so that the compiler masks EXEC itself (lanes 32..63 skip the loop)
  constexpr bool HALF = TEST == T_DPP_HALF_EXEC || TEST == T_FMA_HALF_EXEC;
  t0 = t1 = 0;
  if (!HALF || (threadIdx.x & 63) < 32) {
  STAMP(t0);
  for (int r = 0; r < reps; ++r) {
    if (TEST == T_DPP || TEST == T_DPP_HALF_EXEC)
      asm volatile(DPP16 DPP16 : ACC16 : "v"(b), "v"(c));
    if (TEST == T_DPP_ROWMASK)
      asm volatile(DPPH16 DPPH16 : ACC16 : "v"(b), "v"(c));
    if (TEST == T_FMA || TEST == T_FMA_HALF_EXEC)
      asm volatile(FMA16 FMA16 : ACC16 : "v"(b), "v"(c));
    if (TEST == T_SWAP) // 32 swaps
      asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1\n"
                   : "+v"(w0), "+v"(w1));
    if (TEST == T_MOV32) { // 32 independent 32-bit moves
      asm volatile("v_mov_b32 %0, %16\n v_mov_b32 %1, %16\n v_mov_b32 %2, %16\n v_mov_b32 %3, %16\n v_mov_b32 %4, %16\n v_mov_b32 %5, %16\n v_mov_b32 %6, %16\n v_mov_b32 %7, %16\n"
                   "v_mov_b32 %8, %16\n v_mov_b32 %9, %16\n v_mov_b32 %10, %16\n v_mov_b32 %11, %16\n v_mov_b32 %12, %16\n v_mov_b32 %13, %16\n v_mov_b32 %14, %16\n v_mov_b32 %15, %16\n"
                   "v_mov_b32 %0, %17\n v_mov_b32 %1, %17\n v_mov_b32 %2, %17\n v_mov_b32 %3, %17\n v_mov_b32 %4, %17\n v_mov_b32 %5, %17\n v_mov_b32 %6, %17\n v_mov_b32 %7, %17\n"
                   "v_mov_b32 %8, %17\n v_mov_b32 %9, %17\n v_mov_b32 %10, %17\n v_mov_b32 %11, %17\n v_mov_b32 %12, %17\n v_mov_b32 %13, %17\n v_mov_b32 %14, %17\n v_mov_b32 %15, %17\n"
                   : "+v"(iv[0]), "+v"(iv[1]), "+v"(iv[2]), "+v"(iv[3]), "+v"(iv[4]), "+v"(iv[5]), "+v"(iv[6]), "+v"(iv[7]),
                     "+v"(iv[8]), "+v"(iv[9]), "+v"(iv[10]), "+v"(iv[11]), "+v"(iv[12]), "+v"(iv[13]), "+v"(iv[14]), "+v"(iv[15])
                   : "v"(w0), "v"(w1));
    }
    if (TEST == T_MFMA) { // 8 MFMAs on two independent accumulators
      for (int q = 0; q < 4; ++q) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, acc2, 0, 0, 0);
      }
    }
    if (TEST == T_MFMA_DPP) { // 2 MFMAs + 32 DPP FMAs, independent of each other
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
      asm volatile(DPP16 : ACC16 : "v"(b), "v"(c));
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, acc2, 0, 0, 0);
      asm volatile(DPP16 : ACC16 : "v"(b), "v"(c));
    }
    if (TEST == T_DPP_MOV) { // 32 DPP FMAs + 16 32-bit moves interleaved 2:1
      asm volatile(D(0, 0) D(1, 1) "v_mov_b32 %18, %20\n" D(2, 2) D(3, 3) "v_mov_b32 %19, %20\n" D(4, 4) D(5, 5) "v_mov_b32 %18, %20\n" D(6, 6) D(7, 7) "v_mov_b32 %19, %20\n"
                   D(8, 8) D(9, 9) "v_mov_b32 %18, %20\n" D(10, 10) D(11, 11) "v_mov_b32 %19, %20\n" D(12, 12) D(13, 13) "v_mov_b32 %18, %20\n" D(14, 14) D(15, 15) "v_mov_b32 %19, %20\n"
                   D(0, 0) D(1, 1) "v_mov_b32 %18, %20\n" D(2, 2) D(3, 3) "v_mov_b32 %19, %20\n" D(4, 4) D(5, 5) "v_mov_b32 %18, %20\n" D(6, 6) D(7, 7) "v_mov_b32 %19, %20\n"
                   D(8, 8) D(9, 9) "v_mov_b32 %18, %20\n" D(10, 10) D(11, 11) "v_mov_b32 %19, %20\n" D(12, 12) D(13, 13) "v_mov_b32 %18, %20\n" D(14, 14) D(15, 15) "v_mov_b32 %19, %20\n"
                   : ACC16, "+v"(b), "+v"(c), "+v"(m0), "+v"(m1) : "v"(w0));
    }
  }
  STAMP(t1);
  }
  double s = acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[1] + acc2[2] + acc2[3] + (double)(w0 + w1 + m0 + m1);
  for (int i = 0; i < 16; ++i)
    s += a[i] + (double)iv[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int TEST>
void run(const char *name, int instr_per_rep, const double *in, double *out, unsigned long long *cyc) {
  const int reps = 400;
  for (int wps : {1, 2, 4}) { // wavefronts per SIMD: one workgroup of 256 * wps threads per CU
    const int threads = 256 * wps, blocks = 256;
    for (int it = 0; it < 2; ++it)
      hipLaunchKernelGGL(k<TEST>, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks * threads / 64);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    // cycles the SIMD spends per instruction of ONE wavefront, and per instruction overall
    printf("%-44s %d wave/SIMD: %7.2f cyc/instr per wave  -> %6.2f cyc/instr per SIMD\n", name, wps,
           med / (reps * (double)instr_per_rep), med / (reps * (double)instr_per_rep) / wps);
    fflush(stdout);
  }
}

int main() {
  double *in, *out;
  unsigned long long *cyc;
  hipMalloc(&in, 64 * 17 * 8);
  hipMalloc(&out, 256 * 1024 * 8);
  hipMalloc(&cyc, 256 * 16 * 8);
  std::vector<double> h(64 * 17, 1.0000001);
  hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  run<T_DPP>("v_fmac_f64_dpp row_newbcast", 32, in, out, cyc);
  run<T_DPP_HALF_EXEC>("v_fmac_f64_dpp, EXEC = lanes 0..31", 32, in, out, cyc);
  run<T_DPP_ROWMASK>("v_fmac_f64_dpp, row_mask:0x3", 32, in, out, cyc);
  run<T_FMA>("v_fmac_f64", 32, in, out, cyc);
  run<T_FMA_HALF_EXEC>("v_fmac_f64, EXEC = lanes 0..31", 32, in, out, cyc);
  run<T_SWAP>("v_permlane32_swap", 32, in, out, cyc);
  run<T_MOV32>("v_mov_b32", 32, in, out, cyc);
  run<T_MFMA>("v_mfma_f64_16x16x4_f64", 8, in, out, cyc);
  run<T_MFMA_DPP>("2 mfma_f64 + 32 dpp fma (per 34 instr)", 34, in, out, cyc);
  run<T_DPP_MOV>("32 dpp fma + 16 v_mov_b32 (per 48 instr)", 48, in, out, cyc);
  return 0;
}
