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
  // direct peer copies: one stream per (source rank, destination rank) pair, on the source device
  bool peer_ok = false;
  std::vector<hipStream_t> pair_streams;
  // per rank: work of streams[i] / of done_streams[i] at call time; per pair: copy done
  std::vector<hipEvent_t> ready, ready_done, landed;
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
  // peer access for the direct-copy exchange (best effort: the RCCL paths do not need it)
  int prev = -1;
  (void)hipGetDevice(&prev);
  g->peer_ok = true;
  for (int i = 0; i < ndev && g->peer_ok; ++i) {
    if (hipSetDevice(g->devices[i]) != hipSuccess) {
      g->peer_ok = false;
      break;
    }
    for (int j = 0; j < ndev; ++j) {
      if (j == i)
        continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->devices[i], g->devices[j]) != hipSuccess || !can) {
        g->peer_ok = false;
        break;
      }
      const hipError_t e = hipDeviceEnablePeerAccess(g->devices[j], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        g->peer_ok = false;
      (void)hipGetLastError();
    }
  }
  if (g->peer_ok) {
    g->pair_streams.assign((size_t)ndev * ndev, nullptr);
    g->landed.assign((size_t)ndev * ndev, nullptr);
    g->ready.assign((size_t)ndev, nullptr);
    g->ready_done.assign((size_t)ndev, nullptr);
    for (int i = 0; i < ndev && g->peer_ok; ++i) {
      g->peer_ok = hipSetDevice(g->devices[i]) == hipSuccess &&
                   hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&g->ready_done[i], hipEventDisableTiming) == hipSuccess;
      for (int j = 0; j < ndev && g->peer_ok; ++j)
        g->peer_ok = hipStreamCreateWithFlags(&g->pair_streams[(size_t)i * ndev + j], hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&g->landed[(size_t)i * ndev + j], hipEventDisableTiming) == hipSuccess;
    }
  }
  if (prev >= 0)
    (void)hipSetDevice(prev);
  *out = g;
  return SIP_LQR_OK;
}

void sip_lqr_group_destroy(sip_lqr_group *g) {
  if (g == nullptr)
    return;
  for (hipStream_t s : g->pair_streams)
    if (s != nullptr)
      (void)hipStreamDestroy(s);
  for (hipEvent_t e : g->ready)
    if (e != nullptr)
      (void)hipEventDestroy(e);
  for (hipEvent_t e : g->ready_done)
    if (e != nullptr)
      (void)hipEventDestroy(e);
  for (hipEvent_t e : g->landed)
    if (e != nullptr)
      (void)hipEventDestroy(e);
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
  if (plans[0] == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
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


int sip_lqr_gains_chunk_range(const sip_lqr_plan *plan, int chunk, int num_chunks, int64_t *first_problem,
                              int64_t *num_problems) {
  const int64_t batch = sip_lqr_plan_batch(plan);
  if (plan == nullptr || num_chunks < 1 || num_chunks > batch || chunk < 0 || chunk >= num_chunks)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const int64_t base = batch / num_chunks, rem = batch % num_chunks; // the first `rem` chunks hold one more
  const int64_t lo = chunk * base + (chunk < rem ? chunk : rem);
  if (first_problem)
    *first_problem = lo;
  if (num_problems)
    *num_problems = base + (chunk < rem ? 1 : 0);
  return SIP_LQR_OK;
}

size_t sip_lqr_gains_chunk_offset(const sip_lqr_plan *plan, int nranks, int rank, int chunk, int num_chunks) {
  int64_t lo = 0, count = 0;
  if (nranks < 1 || rank < 0 || rank >= nranks ||
      sip_lqr_gains_chunk_range(plan, chunk, num_chunks, &lo, &count) != SIP_LQR_OK)
    return (size_t)-1;
  const size_t per_problem = sip_lqr_gains_len(plan) * sip_lqr_scalar_bytes(plan);
  return ((size_t)nranks * (size_t)lo + (size_t)rank * (size_t)count) * per_problem;
}

int sip_lqr_all_gather_gains_chunk(const sip_lqr_plan *plan, void *nccl_comm, int nranks, const void *d_gains,
                                   void *d_all_gains, int chunk, int num_chunks, void *stream) {
  int64_t lo = 0, count = 0;
  if (nccl_comm == nullptr || nranks < 1 ||
      sip_lqr_gains_chunk_range(plan, chunk, num_chunks, &lo, &count) != SIP_LQR_OK)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  int comm_ranks = 0; // the chunk's place in the gathered buffer is derived from nranks: it must be the communicator's
  if (ncclCommCount((ncclComm_t)nccl_comm, &comm_ranks) != ncclSuccess || comm_ranks != nranks)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t per_problem = sip_lqr_gains_len(plan) * sip_lqr_scalar_bytes(plan);
  if (per_problem == 0 || count == 0)
    return SIP_LQR_OK;
  if (d_gains == nullptr || d_all_gains == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const ncclResult_t r =
      ncclAllGather((const char *)d_gains + (size_t)lo * per_problem,
                    (char *)d_all_gains + sip_lqr_gains_chunk_offset(plan, nranks, 0, chunk, num_chunks),
                    (size_t)count * per_problem, ncclChar, (ncclComm_t)nccl_comm, (hipStream_t)stream);
  return r == ncclSuccess ? SIP_LQR_OK : fail("sip_lqr_all_gather_gains_chunk", r);
}

int sip_lqr_group_all_gather_gains_chunk(sip_lqr_group *g, const sip_lqr_plan *const *plans,
                                         const void *const *d_gains, void *const *d_all_gains, int chunk,
                                         int num_chunks, void *const *streams) {
  if (g == nullptr || plans == nullptr || d_gains == nullptr || d_all_gains == nullptr || streams == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const int n = (int)g->comms.size();
  int64_t lo = 0, count = 0;
  if (plans[0] == nullptr || sip_lqr_gains_chunk_range(plans[0], chunk, num_chunks, &lo, &count) != SIP_LQR_OK)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t bytes = sip_lqr_gains_bytes(plans[0]);
  for (int i = 0; i < n; ++i)
    if (plans[i] == nullptr || sip_lqr_gains_bytes(plans[i]) != bytes || (bytes > 0 && (!d_gains[i] || !d_all_gains[i])))
      return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t per_problem = sip_lqr_gains_len(plans[0]) * sip_lqr_scalar_bytes(plans[0]);
  if (per_problem == 0 || count == 0)
    return SIP_LQR_OK;
  const size_t out_at = sip_lqr_gains_chunk_offset(plans[0], n, 0, chunk, num_chunks);
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  ncclResult_t r = ncclGroupStart();
  bool hip_failed = false;
  for (int i = 0; i < n && r == ncclSuccess && !hip_failed; ++i) { // no early return inside the group
    if (hipSetDevice(g->devices[i]) != hipSuccess) {
      hip_failed = true;
      break;
    }
    r = ncclAllGather((const char *)d_gains[i] + (size_t)lo * per_problem, (char *)d_all_gains[i] + out_at,
                      (size_t)count * per_problem, ncclChar, g->comms[i], (hipStream_t)streams[i]);
  }
  const ncclResult_t e = ncclGroupEnd();
  if (prev_device >= 0)
    (void)hipSetDevice(prev_device);
  if (hip_failed)
    return SIP_LQR_ERR_HIP;
  if (r != ncclSuccess)
    return fail("sip_lqr_group_all_gather_gains_chunk(ncclAllGather)", r);
  return e == ncclSuccess ? SIP_LQR_OK : fail("sip_lqr_group_all_gather_gains_chunk(ncclGroupEnd)", e);
}

int sip_lqr_group_all_gather_gains_p2p_chunk(sip_lqr_group *g, const sip_lqr_plan *const *plans,
                                             const void *const *d_gains, void *const *d_all_gains, int chunk,
                                             int num_chunks, void *const *streams, void *const *done_streams) {
  if (g == nullptr || plans == nullptr || d_gains == nullptr || d_all_gains == nullptr || streams == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (!g->peer_ok)
    return SIP_LQR_ERR_UNSUPPORTED;
  const int n = (int)g->devices.size();
  int64_t lo = 0, count = 0;
  if (plans[0] == nullptr || sip_lqr_gains_chunk_range(plans[0], chunk, num_chunks, &lo, &count) != SIP_LQR_OK)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t bytes = sip_lqr_gains_bytes(plans[0]);
  for (int i = 0; i < n; ++i)
    if (plans[i] == nullptr || sip_lqr_gains_bytes(plans[i]) != bytes || (bytes > 0 && (!d_gains[i] || !d_all_gains[i])))
      return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t per_problem = sip_lqr_gains_len(plans[0]) * sip_lqr_scalar_bytes(plans[0]);
  const size_t chunk_bytes = (size_t)count * per_problem;
  if (chunk_bytes == 0)
    return SIP_LQR_OK;
  if (done_streams == nullptr)
    done_streams = streams;
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  hipError_t e = hipSuccess;
  // What rank i has enqueued so far on streams[i] produces its chunk; what it has enqueued on
  // done_streams[i] may still read the gathered buffer the copies are about to overwrite.
  for (int i = 0; i < n && e == hipSuccess; ++i) {
    e = hipSetDevice(g->devices[i]);
    if (e == hipSuccess)
      e = hipEventRecord(g->ready[i], (hipStream_t)streams[i]);
    if (e == hipSuccess)
      e = hipEventRecord(g->ready_done[i], (hipStream_t)done_streams[i]);
  }
  for (int i = 0; i < n && e == hipSuccess; ++i) { // source rank i: one copy per destination, all at once
    e = hipSetDevice(g->devices[i]);
    const char *src = (const char *)d_gains[i] + (size_t)lo * per_problem;
    for (int j = 0; j < n && e == hipSuccess; ++j) {
      const hipStream_t ps = g->pair_streams[(size_t)i * n + j];
      char *dst = (char *)d_all_gains[j] + sip_lqr_gains_chunk_offset(plans[0], n, i, chunk, num_chunks);
      e = hipStreamWaitEvent(ps, g->ready[i], 0);
      if (e == hipSuccess) // the destination may still be read / written by what rank j has enqueued
        e = hipStreamWaitEvent(ps, g->ready_done[j], 0);
      if (e == hipSuccess && j != i)
        e = hipStreamWaitEvent(ps, g->ready[j], 0);
      if (e == hipSuccess)
        e = hipMemcpyPeerAsync(dst, g->devices[j], src, g->devices[i], chunk_bytes, ps);
      if (e == hipSuccess)
        e = hipEventRecord(g->landed[(size_t)i * n + j], ps);
    }
  }
  // done_streams[r] continues once everything has landed ON r (column r of the event matrix: the
  // gathered chunk is complete) and every copy OUT of r has read its source (row r: d_gains[r] may be
  // rewritten).  With done_streams == streams that is ncclAllGather's stream semantics: both buffers
  // are safe to reuse behind the call.
  for (int r = 0; r < n && e == hipSuccess; ++r) {
    e = hipSetDevice(g->devices[r]);
    for (int q = 0; q < n && e == hipSuccess; ++q) {
      e = hipStreamWaitEvent((hipStream_t)done_streams[r], g->landed[(size_t)q * n + r], 0);
      if (e == hipSuccess && q != r)
        e = hipStreamWaitEvent((hipStream_t)done_streams[r], g->landed[(size_t)r * n + q], 0);
    }
  }
  if (prev_device >= 0)
    (void)hipSetDevice(prev_device);
  if (e != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_group_all_gather_gains_p2p: %s\n", hipGetErrorString(e));
    return SIP_LQR_ERR_HIP;
  }
  return SIP_LQR_OK;
}

int sip_lqr_group_all_gather_gains_p2p(sip_lqr_group *g, const sip_lqr_plan *const *plans,
                                       const void *const *d_gains, void *const *d_all_gains,
                                       void *const *streams) {
  // one chunk of one: chunk-major and rank-major layouts coincide
  return sip_lqr_group_all_gather_gains_p2p_chunk(g, plans, d_gains, d_all_gains, 0, 1, streams, nullptr);
}

} // extern "C"
