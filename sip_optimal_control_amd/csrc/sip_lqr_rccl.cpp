// sip_lqr_rccl.cpp -- include/sip_lqr_amd_rccl.h over RCCL (ncclAllGather on xGMI).
#include "../../include/sip_lqr_amd_rccl.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <new>
#include <vector>

struct sip_lqr_group {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
};

namespace {
int fail(const char *what, ncclResult_t r) {
  std::fprintf(stderr, "%s: %s\n", what, ncclGetErrorString(r));
  return SIP_LQR_ERR_HIP;
}
} // namespace

extern "C" {

int sip_lqr_group_create(int ndev, const int *devices, sip_lqr_group **out) {
  if (out == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (ndev < 1 || devices == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sip_lqr_group *g = new (std::nothrow) sip_lqr_group;
  if (g == nullptr)
    return SIP_LQR_ERR_ALLOC;
  g->devices.assign(devices, devices + ndev);
  g->comms.assign((size_t)ndev, nullptr);
  const ncclResult_t r = ncclCommInitAll(g->comms.data(), ndev, g->devices.data());
  if (r != ncclSuccess) {
    delete g;
    return fail("sip_lqr_group_create(ncclCommInitAll)", r);
  }
  *out = g;
  return SIP_LQR_OK;
}

void sip_lqr_group_destroy(sip_lqr_group *g) {
  if (g == nullptr)
    return;
  for (ncclComm_t c : g->comms)
    if (c != nullptr)
      (void)ncclCommDestroy(c);
  delete g;
}

int sip_lqr_group_size(const sip_lqr_group *g) { return g ? (int)g->comms.size() : 0; }

int sip_lqr_group_all_gather_gains(sip_lqr_group *g, const sip_lqr_plan *const *plans,
                                   const void *const *d_gains, void *const *d_all_gains,
                                   void *const *streams) {
  if (g == nullptr || plans == nullptr || d_gains == nullptr || d_all_gains == nullptr || streams == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const int n = (int)g->comms.size();
  const size_t bytes = sip_lqr_gains_bytes(plans[0]);
  for (int i = 0; i < n; ++i)
    if (plans[i] == nullptr || sip_lqr_gains_bytes(plans[i]) != bytes || (bytes > 0 && (!d_gains[i] || !d_all_gains[i])))
      return SIP_LQR_ERR_INVALID_ARGUMENT; // equal shards: the batch is block-partitioned evenly
  if (bytes == 0)
    return SIP_LQR_OK;
  int prev_device = -1;
  (void)hipGetDevice(&prev_device); // restored below: the loop visits every rank's device
  ncclResult_t r = ncclGroupStart();
  bool hip_failed = false;
  // No early return between ncclGroupStart and ncclGroupEnd: an open group would swallow every
  // later collective of this thread.
  for (int i = 0; i < n && r == ncclSuccess && !hip_failed; ++i) {
    if (hipSetDevice(g->devices[i]) != hipSuccess) {
      hip_failed = true;
      break;
    }
    r = ncclAllGather(d_gains[i], d_all_gains[i], bytes, ncclChar, g->comms[i], (hipStream_t)streams[i]);
  }
  const ncclResult_t e = ncclGroupEnd();
  if (prev_device >= 0)
    (void)hipSetDevice(prev_device);
  if (hip_failed)
    return SIP_LQR_ERR_HIP;
  if (r != ncclSuccess)
    return fail("sip_lqr_group_all_gather_gains(ncclAllGather)", r);
  return e == ncclSuccess ? SIP_LQR_OK : fail("sip_lqr_group_all_gather_gains(ncclGroupEnd)", e);
}

int sip_lqr_all_gather_gains(const sip_lqr_plan *plan, void *nccl_comm, const void *d_gains, void *d_all_gains,
                             void *stream) {
  if (plan == nullptr || nccl_comm == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t bytes = sip_lqr_gains_bytes(plan);
  if (bytes == 0)
    return SIP_LQR_OK;
  if (d_gains == nullptr || d_all_gains == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const ncclResult_t r = ncclAllGather(d_gains, d_all_gains, bytes, ncclChar, (ncclComm_t)nccl_comm, (hipStream_t)stream);
  return r == ncclSuccess ? SIP_LQR_OK : fail("sip_lqr_all_gather_gains", r);
}

} // extern "C"
