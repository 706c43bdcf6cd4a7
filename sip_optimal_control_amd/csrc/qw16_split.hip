// qw16_split.hip -- the split form of the fused chain sweep (chain_qw16.hpp, SPLIT): the stage blocks of
// `mats` carry [Q | delta | M | R] and the dynamics Jacobians A | B stream from where the model callback
// left them.  Used by the Newton-KKT step (sip_kkt_factor_solve), whose condensation then neither reads
// nor copies A | B (helpers.cpp:365-366 copies them into the LQR inputs).  Compiled in four slices
// (-DSIP_QW16_SPLIT_SLICE=0..3) beside the other translation units.
#include "qw16_launch.hpp"

#ifndef SIP_QW16_SPLIT_SLICE
#define SIP_QW16_SPLIT_SLICE 0
#endif

namespace sipamd {

const SplitEntry *qw16_split_slice_1(int *count);
const SplitEntry *qw16_split_slice_2(int *count);
const SplitEntry *qw16_split_slice_3(int *count);

// the staged shapes of the reference's Newton-KKT benchmark grid (newton_kkt_benchmark.cpp:264-273:
// n in {4, 6, 8}, m in {1, 2, 3, 4}) and n = 12 (the f1 shape of bench.py and its fewer-control relatives)
#if SIP_QW16_SPLIT_SLICE == 0

namespace {
const SplitEntry kSplit[] = {QW16_SPLIT(12, 4), QW16_SPLIT(4, 2), QW16_SPLIT(4, 4), QW16_SPLIT(6, 2)};
}

launch_split_t find_split_launch(int n, int m, int layout) {
  for (const SplitEntry &e : kSplit)
    if (e.n == n && e.m == m && e.layout == layout)
      return e.launch;
#if !defined(SIP_QW16_QUICK) && !defined(SIP_QW16_NO_EXTRA) // tools/ab_build.sh, tools/diag_build.sh link slice 0 alone
  typedef const SplitEntry *(*slice_fn)(int *);
  for (const slice_fn fn : {qw16_split_slice_1, qw16_split_slice_2, qw16_split_slice_3}) {
    int count = 0;
    const SplitEntry *more = fn(&count);
    for (int k = 0; k < count; ++k)
      if (more[k].n == n && more[k].m == m && more[k].layout == layout)
        return more[k].launch;
  }
  // every other staged shape n <= 15, m <= 8: beside its fused kernel in the slices of qw16_extra.hip
  for (const slice_fn fn : {qw16_extra_split_slice_0, qw16_extra_split_slice_1, qw16_extra_split_slice_2,
                            qw16_extra_split_slice_3, qw16_extra_split_slice_4, qw16_extra_split_slice_5,
                            qw16_extra_split_slice_6, qw16_extra_split_slice_7}) {
    int count = 0;
    const SplitEntry *more = fn(&count);
    for (int k = 0; k < count; ++k)
      if (more[k].n == n && more[k].m == m && more[k].layout == layout)
        return more[k].launch;
  }
#endif
  return nullptr;
}

long split_mats_stage(int n, int m, int layout) {
  const bool sym = layout == SIP_LQR_LAYOUT_SYMMETRIC;
  return (sym ? (long)n * (n + 1) / 2 : (long)n * n) + n + (long)n * m + (sym ? (long)m * (m + 1) / 2 : (long)m * m);
}

#else

namespace {
#if SIP_QW16_SPLIT_SLICE == 1
const SplitEntry kSplit[] = {QW16_SPLIT(6, 4), QW16_SPLIT(8, 2), QW16_SPLIT(8, 4), QW16_SPLIT(12, 2)};
#elif SIP_QW16_SPLIT_SLICE == 2
const SplitEntry kSplit[] = {QW16_SPLIT(12, 3), QW16_SPLIT(4, 1), QW16_SPLIT(4, 3), QW16_SPLIT(6, 1), QW16_SPLIT_SYM(12, 4)};
#else
const SplitEntry kSplit[] = {QW16_SPLIT(12, 1), QW16_SPLIT(6, 3), QW16_SPLIT(8, 1), QW16_SPLIT(8, 3), QW16_SPLIT_SYM(8, 4),
                             QW16_SPLIT_SYM(4, 4)};
#endif
} // namespace

#define SIP_SPLIT_SLICE_FN_(S) qw16_split_slice_##S
#define SIP_SPLIT_SLICE_FN(S) SIP_SPLIT_SLICE_FN_(S)
const SplitEntry *SIP_SPLIT_SLICE_FN(SIP_QW16_SPLIT_SLICE)(int *count) {
  *count = (int)(sizeof(kSplit) / sizeof(kSplit[0]));
  return kSplit;
}

#endif

} // namespace sipamd
