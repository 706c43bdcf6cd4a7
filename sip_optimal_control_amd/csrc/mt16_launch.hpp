// mt16_launch.hpp -- launchers of the n = 32 matrix-core chain kernels (chain_mt16.hpp), instantiated
// in chain_mt16.hip (a translation unit of its own: compiled in parallel with the others).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sipamd {

// Same signature as launch_fs_t (qw16_launch.hpp); `mode` is ignored: the kernel always runs the full
// sweep (split factor / solve calls re-run it, see sip_lqr_plan::split_on_fused).
template <typename S, int M>
hipError_t launch_mt16(long batch, int T, const void *mats, const void *vecs, void *sol, void *gains,
                       int32_t *status, void *ws, hipStream_t stream, int mode, void *gfac);

constexpr int kMt16SpillPerNode = 3 * 256 + 32; // Layout::WSN: three tiles of the symmetric W | g

} // namespace sipamd

#define MT16_ENTRY(DT, S, TAG, M)                                                                            \
  { DT, 32, M, "chain_factor_solve_mt16<32," #M ",mfma16x16x4>/" TAG, sipamd::kMt16SpillPerNode,             \
    &sipamd::launch_mt16<S, M> }
