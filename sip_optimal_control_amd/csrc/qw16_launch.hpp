// qw16_launch.hpp -- host-side launcher and table entry of the fused chain kernels, shared by the
// translation units that instantiate them (sip_lqr_amd.hip: the benchmark shapes and the hosts of
// the embedding; qw16_extra.hip, compiled in slices: every other shape n <= 16, m <= 8).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/sip_lqr_amd.h"
#include "chain_mrhs.hpp"
#include "chain_qw16.hpp"

#ifdef SIP_LQR_STAMPS
// Diagnostic build: device buffer of 8 x u64 per wave, set by the tool (sip_lqr_amd.hip).
extern unsigned long long *g_sip_lqr_stamps;
#define SIP_STAMP_PASS , g_sip_lqr_stamps
#else
#define SIP_STAMP_PASS
#endif

namespace sipamd {

// mode: 0 fused factor + solve, 1 factor only (+ the G factors to gfac), 2 solve only
typedef hipError_t (*launch_fs_t)(long batch, int T, const void *mats, const void *vecs, void *sol, void *gains,
                                  int32_t *status, void *ws, hipStream_t stream, int mode, void *gfac);

// LQR::solve for `ncols` right-hand sides in one sweep (chain_mrhs.hpp); columns `col_stride` scalars apart
typedef hipError_t (*launch_mrhs_t)(long batch, int T, const void *mats, const void *vecs_cols, void *sol_cols,
                                    const void *gains, const void *ws, const void *gfac, void *cws,
                                    const int32_t *status, int ncols, long col_stride, hipStream_t stream);
constexpr int kMrhsColumns = 8; // columns one multi-rhs launch carries
#ifndef SIP_MRHS_GROUP
#define SIP_MRHS_GROUP 8 // (measured at (12, 4), 8 columns: groups of 8 / 4 / 2 -> 0.69 / 0.94 / 1.3 ms: the sweep is bandwidth-bound, smaller groups re-fetch the operands)
#endif
constexpr int kMrhsGroup = SIP_MRHS_GROUP; // ... of which one wavefront carries this many (chain_mrhs.hpp)

struct KernelEntry {
  int dtype, n, m;
  const char *name;
  int ws_slot; // scalars of workspace per node
  launch_fs_t launch_fs;
  launch_mrhs_t launch_mrhs; // nullptr: this shape solves several right-hand sides column by column
  int layout;                // SIP_LQR_LAYOUT_* of mats the kernel reads (0: the full squares)
};

template <int N, int M, bool WPACK>
hipError_t launch_mrhs_qw16(long batch, int T, const void *mats, const void *vecs_cols, void *sol_cols,
                            const void *gains, const void *ws, const void *gfac, void *cws, const int32_t *status,
                            int ncols, long col_stride, hipStream_t stream) {
  if (ncols < 1 || ncols > kMrhsColumns)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((chain_solve_mrhs_qw16<N, M, WPACK, kMrhsGroup>),
                     dim3((unsigned)((batch + 3) / 4), (unsigned)((ncols + kMrhsGroup - 1) / kMrhsGroup)), dim3(64), 0, stream, (const double *)mats, (const double *)vecs_cols, (double *)sol_cols,
                     (const double *)gains, (const double *)ws, (const double *)gfac, (double *)cws,
                     (const int *)status, batch, T, ncols, col_stride);
  return hipGetLastError();
}

template <int N, int M, bool STAGED, bool WPACK, bool SYM = false>
hipError_t launch_qw16(long batch, int T, const void *mats, const void *vecs, void *sol, void *gains,
                       int32_t *status, void *ws, hipStream_t stream, int mode, void *gfac) {
  using Cfg = StagedCfg<N, M, WPACK, false, SYM>;
  const unsigned blocks = (unsigned)((batch + 3) / 4);
  const unsigned lds = STAGED ? Cfg::LDS_BYTES : 0;
  if (STAGED) {
    // LDS-DMA moves 16-byte pieces: every base must be 16-byte aligned.
    const uintptr_t bits = (uintptr_t)mats | (uintptr_t)vecs | (uintptr_t)gains | (uintptr_t)ws;
    if (bits & 15)
      return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL((chain_factor_solve_qw16<N, M, STAGED, WPACK, false, SYM>), dim3(blocks), dim3(64), lds, stream,
                     (const double *)mats, (const double *)vecs, (double *)sol, (double *)gains, (double *)ws,
                     (int *)status, batch, T, mode, (double *)gfac, (const double *)nullptr, 0L, 0L SIP_STAMP_PASS);
  return hipGetLastError();
}

// The split form of the fused sweep (sip_lqr_factor_solve_split): mats carries [Q | delta | M | R] per
// stage, A | B stream from `ab` (scalars: ab + p * ab_pstride + i * ab_sstride).  qw16_split.hip
// instantiates it for the staged shapes of the reference's Newton-KKT benchmark grid.
typedef hipError_t (*launch_split_t)(long batch, int T, const void *mats, const void *ab, long ab_pstride,
                                     long ab_sstride, const void *vecs, void *sol, void *gains, int32_t *status,
                                     void *ws, hipStream_t stream);
template <int N, int M, bool SYM = false>
hipError_t launch_qw16_split(long batch, int T, const void *mats, const void *ab, long ab_pstride, long ab_sstride,
                             const void *vecs, void *sol, void *gains, int32_t *status, void *ws,
                             hipStream_t stream) {
  using Cfg = StagedCfg<N, M, true, true, SYM>;
  // 16-byte aligned arrays of this library's own layouts; A | B -- the caller's -- at any 8-byte aligned place and
  // strides: a stage's A | B is a whole number of 16-byte pieces (AB even, StagedCfg) and the LDS-DMA of gfx950 copies
  // pieces exactly from sources that are only 8-byte aligned (tools/ubench/lds_dma_align.hip), as the odd-length
  // stage blocks of `mats` already rely on
  const uintptr_t bits = (uintptr_t)mats | (uintptr_t)vecs | (uintptr_t)gains | (uintptr_t)ws;
  if ((bits & 15) || ((uintptr_t)ab & 7))
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((chain_factor_solve_qw16<N, M, true, true, true, SYM>), dim3((unsigned)((batch + 3) / 4)), dim3(64),
                     Cfg::LDS_BYTES, stream, (const double *)mats, (const double *)vecs, (double *)sol,
                     (double *)gains, (double *)ws, (int *)status, batch, T, 0, (double *)nullptr,
                     (const double *)ab, ab_pstride, ab_sstride SIP_STAMP_PASS);
  return hipGetLastError();
}
struct SplitEntry {
  int n, m;
  launch_split_t launch;
  int layout; // SIP_LQR_LAYOUT_* of the [Q | delta | M | R] blocks
};
#define QW16_SPLIT(N, M) {N, M, &sipamd::launch_qw16_split<N, M>, SIP_LQR_LAYOUT_FULL}
#define QW16_SPLIT_SYM(N, M) {N, M, &sipamd::launch_qw16_split<N, M, true>, SIP_LQR_LAYOUT_SYMMETRIC}
launch_split_t find_split_launch(int n, int m, int layout = 0); // qw16_split.hip; nullptr: no split kernel for the shape / layout
long split_mats_stage(int n, int m, int layout = 0);            // scalars of [Q | delta | M | R]

// The shapes n <= 16, m <= 8 that sip_lqr_amd.hip does not instantiate itself live in eight slices
// of qw16_extra.hip (compiled in parallel): slice s defines qw16_extra_slice_<s>.
constexpr int kQw16ExtraSlices = 8;
#define SIP_QW16_DECLARE_SLICE(S)                                \
  const KernelEntry *qw16_extra_slice_##S(int *count);          \
  const SplitEntry *qw16_extra_split_slice_##S(int *count); // the split kernels of the slice's staged shapes
SIP_QW16_DECLARE_SLICE(0) SIP_QW16_DECLARE_SLICE(1) SIP_QW16_DECLARE_SLICE(2) SIP_QW16_DECLARE_SLICE(3)
SIP_QW16_DECLARE_SLICE(4) SIP_QW16_DECLARE_SLICE(5) SIP_QW16_DECLARE_SLICE(6) SIP_QW16_DECLARE_SLICE(7)
#undef SIP_QW16_DECLARE_SLICE

} // namespace sipamd

// symmetric-packed layout (SIP_LQR_LAYOUT_SYMMETRIC): Q and R as packed lower triangles; staged kernels, even n and m
#define QW16_STAGED_SYM(N, M)                                                                                 \
  { SIP_LQR_F64, N, M, "chain_factor_solve_qw16<" #N "," #M ",staged,sym>/f64",                               \
    sipamd::StagedCfg<N, M, true, false, true>::WSN, &sipamd::launch_qw16<N, M, true, true, true>, nullptr,   \
    SIP_LQR_LAYOUT_SYMMETRIC }
// direct: every lane loads its columns from global memory (any N <= 16)
#define QW16_DIRECT(N, M)                                                                                     \
  { SIP_LQR_F64, N, M, "chain_factor_solve_qw16<" #N "," #M ",direct>/f64",                                   \
    sipamd::StagedCfg<N, M, false>::WSN, &sipamd::launch_qw16<N, M, false, false> }
// staged: LDS-DMA double buffering + packed symmetric S spill (N, M even, N <= 14)
#define QW16_STAGED(N, M)                                                                                     \
  { SIP_LQR_F64, N, M, "chain_factor_solve_qw16<" #N "," #M ",staged>/f64",                                   \
    sipamd::StagedCfg<N, M, true>::WSN, &sipamd::launch_qw16<N, M, true, true> }
// ... with the multi-right-hand-side solve kernel of chain_mrhs.hpp (the shapes of the reference's
// Newton-KKT benchmark grid, where the theta Schur complement solves p columns per factorization)
#define QW16_STAGED_MR(N, M)                                                                                  \
  { SIP_LQR_F64, N, M, "chain_factor_solve_qw16<" #N "," #M ",staged>/f64",                                   \
    sipamd::StagedCfg<N, M, true>::WSN, &sipamd::launch_qw16<N, M, true, true>,                               \
    &sipamd::launch_mrhs_qw16<N, M, true> }
#define QW16_DIRECT_MR(N, M)                                                                                  \
  { SIP_LQR_F64, N, M, "chain_factor_solve_qw16<" #N "," #M ",direct>/f64",                                   \
    sipamd::StagedCfg<N, M, false>::WSN, &sipamd::launch_qw16<N, M, false, false>,                            \
    &sipamd::launch_mrhs_qw16<N, M, false> }
