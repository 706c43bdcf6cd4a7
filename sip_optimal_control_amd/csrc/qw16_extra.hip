// qw16_extra.hip -- one slice (-DSIP_QW16_SLICE=0..7) of the fused fp64 chain kernels for the
// shapes n <= 16, m <= 8 outside the table of sip_lqr_amd.hip, so that every such shape runs on an
// exact instantiation (no embedding, no general engine).  The shape list is generated
// (gen_qw16_extra.py -> qw16_extra_shapes_gen.hpp); the slices compile in parallel.
#include "qw16_launch.hpp"
#include "qw16_extra_shapes_gen.hpp"

#ifdef SIP_QW16_FORCE_DIRECT // diagnostic: the direct-load variant of every shape of the slice
#undef QW16_STAGED
#define QW16_STAGED(N, M) QW16_DIRECT(N, M)
#endif
#ifndef SIP_QW16_SLICE
#error "compile with -DSIP_QW16_SLICE=<0..7>"
#endif
#define SIP_CAT2(a, b) a##b
#define SIP_CAT(a, b) SIP_CAT2(a, b)

namespace sipamd {
namespace {
const KernelEntry kSlice[] = {QW16_SLICE_ENTRIES};
}
const KernelEntry *SIP_CAT(qw16_extra_slice_, SIP_QW16_SLICE)(int *count) {
  *count = (int)(sizeof(kSlice) / sizeof(kSlice[0]));
  return kSlice;
}
// the split form (A | B read where the model callback left them, qw16_split.hip) of the slice's share of the
// staged shapes: with these, sip_lqr_factor_solve_split covers every shape n <= 15, m <= 8
namespace {
const SplitEntry kSplitSlice[] = {QW16_SLICE_SPLIT_ENTRIES};
}
const SplitEntry *SIP_CAT(qw16_extra_split_slice_, SIP_QW16_SLICE)(int *count) {
  *count = (int)(sizeof(kSplitSlice) / sizeof(kSplitSlice[0]));
  return kSplitSlice;
}
} // namespace sipamd
