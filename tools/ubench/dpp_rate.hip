// Micro-benchmark: issue rate / latency of fp64 FMA vs v_fmac_f64_dpp row_newbcast
// on gfx950, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 dpp_rate.hip -o dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")

__global__ void k(const double* in, double* out, unsigned long long* cyc) {
  double a[16], b = in[threadIdx.x + 1], acc[16];
  for (int i = 0; i < 16; i++) { a[i] = in[threadIdx.x * 16 + i]; acc[i] = 0; }
  unsigned long long t0, t1, t2, t3, t4, t5, t6, t7;
  // warm
  asm volatile("s_nop 4");
  STAMP(t0);
  // (A) 16 independent accumulators, plain fma: 16*8 = 128 instr
#define A16(OP) \
  asm volatile(OP : "+v"(acc[0]),"+v"(acc[1]),"+v"(acc[2]),"+v"(acc[3]),"+v"(acc[4]),"+v"(acc[5]),"+v"(acc[6]),"+v"(acc[7]), \
     "+v"(acc[8]),"+v"(acc[9]),"+v"(acc[10]),"+v"(acc[11]),"+v"(acc[12]),"+v"(acc[13]),"+v"(acc[14]),"+v"(acc[15]) \
     : "v"(a[0]),"v"(a[1]),"v"(a[2]),"v"(a[3]),"v"(a[4]),"v"(a[5]),"v"(a[6]),"v"(a[7]),"v"(a[8]),"v"(a[9]),"v"(a[10]),"v"(a[11]),"v"(a[12]),"v"(a[13]),"v"(a[14]),"v"(a[15]),"v"(b));
#define FMA16 \
  "v_fmac_f64 %0, %16, %32\n v_fmac_f64 %1, %17, %32\n v_fmac_f64 %2, %18, %32\n v_fmac_f64 %3, %19, %32\n" \
  "v_fmac_f64 %4, %20, %32\n v_fmac_f64 %5, %21, %32\n v_fmac_f64 %6, %22, %32\n v_fmac_f64 %7, %23, %32\n" \
  "v_fmac_f64 %8, %24, %32\n v_fmac_f64 %9, %25, %32\n v_fmac_f64 %10, %26, %32\n v_fmac_f64 %11, %27, %32\n" \
  "v_fmac_f64 %12, %28, %32\n v_fmac_f64 %13, %29, %32\n v_fmac_f64 %14, %30, %32\n v_fmac_f64 %15, %31, %32\n"
#define D(i,j,l) "v_fmac_f64_dpp %" #i ", %" #j ", %32 row_newbcast:" #l " row_mask:0xf bank_mask:0xf\n"
#define DPP16 D(0,16,0) D(1,17,1) D(2,18,2) D(3,19,3) D(4,20,4) D(5,21,5) D(6,22,6) D(7,23,7) D(8,24,8) D(9,25,9) D(10,26,10) D(11,27,11) D(12,28,12) D(13,29,13) D(14,30,14) D(15,31,15)
  REP8(A16(FMA16))
  STAMP(t1);
  REP8(A16(DPP16))
  STAMP(t2);
  // (C) dependent chain fma: 64 instr on one accumulator
#define CH(OP) asm volatile(OP : "+v"(acc[0]) : "v"(a[0]), "v"(b));
#define C8F "v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2\n"
#define C8D "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
  REP8(CH(C8F))
  STAMP(t3);
  REP8(CH(C8D))
  STAMP(t4);
  // (E) dependent chain where the DPP source is the freshly written accumulator (needs s_nop 1)
#define C8E "s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
  REP8(CH(C8E))
  STAMP(t5);
  // (F) rsq chain 16, mul chain
  double r = a[1];
  asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n"
               "v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n v_rsq_f64 %0, %0\n" : "+v"(r));
  STAMP(t6);
  // (G) 64 x s_nop 1 + independent dpp pairs: cost of s_nop
  asm volatile("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n"
               "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
  STAMP(t7);
  double s = r;
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    unsigned long long* o = cyc + blockIdx.x * 8;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t6 - t5; o[6] = t7 - t6;
  }
}

int main() {
  int blocks = 1024;
  double *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 64 * 17 * 8); hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8 * 8);
  std::vector<double> h(64 * 17, 1.0000001);
  hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, in, out, cyc);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(blocks * 8);
  hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"128 indep v_fmac_f64", "128 indep v_fmac_f64_dpp", "64 dep v_fmac_f64", "64 dep v_fmac_f64_dpp(src const)", "64 dep (s_nop1 + dpp on acc)", "16 dep v_rsq_f64", "16 s_nop1 + 16 s_nop0"};
  int counts[] = {128, 128, 64, 64, 64, 16, 32};
  for (int j = 0; j < 7; j++) {
    std::vector<unsigned long long> v;
    for (int b = 0; b < blocks; b++) v.push_back(c[b * 8 + j]);
    std::sort(v.begin(), v.end());
    printf("%-36s median %6llu cycles  -> %.2f / instr (min %llu)\n", names[j], v[blocks / 2], (double)v[blocks / 2] / counts[j], v[0]);
  }
  return 0;
}
