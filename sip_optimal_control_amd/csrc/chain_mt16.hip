// chain_mt16.hip -- instantiations of the n = 32 matrix-core chain kernel (chain_mt16.hpp).
#include "chain_mt16.hpp"
#include "mt16_launch.hpp"

namespace sipamd {

template <typename S, int M>
hipError_t launch_mt16(long batch, int T, const void *mats, const void *vecs, void *sol, void *gains,
                       int32_t *status, void *ws, hipStream_t stream, int /*mode: always the full sweep*/,
                       void * /*gfac*/) {
  // 16-byte loads of the stage blocks (fp32, even m) and of the W dump: bases must be 16-byte aligned
  if (((uintptr_t)mats | (uintptr_t)ws) & 15)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((mt16::chain_factor_solve_mt16<S, M>), dim3((unsigned)batch), dim3(64), 0, stream,
                     (const S *)mats, (const S *)vecs, (S *)sol, (S *)gains, (S *)ws, (int *)status, batch, T);
  return hipGetLastError();
}

static_assert(mt16::Layout<float, 8>::WSN == kMt16SpillPerNode && mt16::Layout<double, 4>::WSN == kMt16SpillPerNode,
              "the dispatch table sizes the workspace by kMt16SpillPerNode");

#define SIP_MT16_INSTANTIATE(S, M)                                                                           \
  template hipError_t launch_mt16<S, M>(long, int, const void *, const void *, void *, void *, int32_t *,    \
                                        void *, hipStream_t, int, void *);
SIP_MT16_INSTANTIATE(float, 8)
SIP_MT16_INSTANTIATE(float, 4)
SIP_MT16_INSTANTIATE(double, 8)
SIP_MT16_INSTANTIATE(double, 4)

} // namespace sipamd
