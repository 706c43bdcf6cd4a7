// tree_qw16.hip -- size classes of the fused tree kernel (tree_qw16.hpp) and their launcher; its
// own translation unit so that it compiles beside the others.
#include "tree_qw16_launch.hpp"

namespace sipamd {

namespace {
template <int N, int M, bool EXPORT>
hipError_t launch(const TreeSchedule &ts, const double *input, double *output, double *work, double *pgains,
                  double *spill, int32_t *status, long batch, hipStream_t s) {
  if (EXPORT && work == nullptr)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((tree_factor_solve_qw16<N, M, EXPORT>), dim3((unsigned)((batch + 3) / 4)), dim3(64), 0, s, ts,
                     input, output, work, pgains, spill, (int *)status, batch);
  return hipGetLastError();
}
#define TREE_CLASS(N, M) {N, M, "tree_factor_solve_qw16<" #N "," #M ">/f64", &launch<N, M, false>, &launch<N, M, true>}
// sorted by cost: the first class that holds the largest node and the largest control wins
// ((9, 3): the reference's variable-shape benchmark family at base dimension 8 -- states 7..9, controls 1..3,
// benchmarks/lqr_benchmark.cpp:209-310 -- padded to (10, 4) before round 3)
const TreeClass kClasses[] = {TREE_CLASS(4, 2),  TREE_CLASS(6, 3),  TREE_CLASS(8, 4),  TREE_CLASS(9, 3), TREE_CLASS(10, 4),
                              TREE_CLASS(12, 4), TREE_CLASS(15, 4), TREE_CLASS(15, 8)};
#undef TREE_CLASS
} // namespace

const TreeClass *find_tree_class(int max_n, int max_m) {
  for (const TreeClass &k : kClasses)
    if (k.n >= max_n && k.m >= max_m)
      return &k;
  return nullptr;
}

} // namespace sipamd

#ifdef SIP_TREE_STAMPS
// diagnostic build: read (and clear) the per-segment cycle sums of tree_factor_solve_qw16 (tools/tree_stamps.py)
extern "C" void sip_lqr_tree_debug_segments(unsigned long long *out8) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(sipamd::g_tree_seg), 8 * sizeof(unsigned long long));
  unsigned long long zero[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sipamd::g_tree_seg), zero, sizeof(zero));
}
#endif
