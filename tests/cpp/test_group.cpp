// test_group.cpp -- include/sip_lqr_amd_rccl.h on the devices that are present (one on the test box):
// a sweep per device, then the all-gather of the gains through sip_lqr_group_all_gather_gains and,
// with the same communicator handed over as the caller's, sip_lqr_all_gather_gains.
#include "sip_lqr_amd_rccl.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <vector>

static int g_failures = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      ++g_failures;                                                         \
      std::printf("  CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
    }                                                                       \
  } while (0)

int main() {
  int ndev = 0;
  CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev >= 1);
  if (ndev > 4)
    ndev = 4;
  std::vector<int> devices(ndev);
  for (int i = 0; i < ndev; ++i)
    devices[i] = i;
  sip_lqr_group *group = nullptr;
  CHECK(sip_lqr_group_create(ndev, devices.data(), &group) == SIP_LQR_OK && group != nullptr);
  CHECK(sip_lqr_group_size(group) == ndev);
  sip_lqr_group *none = nullptr;
  CHECK(sip_lqr_group_create(0, devices.data(), &none) == SIP_LQR_ERR_INVALID_ARGUMENT && none == nullptr);

  const int T = 6, n = 4, m = 2;
  const int64_t batch = 5;
  std::vector<sip_lqr_plan *> plans(ndev, nullptr);
  std::vector<void *> mats(ndev), vecs(ndev), sol(ndev), gains(ndev), all(ndev), ws(ndev), streams(ndev, nullptr);
  std::vector<int32_t *> status(ndev);
  std::vector<std::vector<double>> host_gains(ndev);
  for (int r = 0; r < ndev; ++r) {
    CHECK(hipSetDevice(r) == hipSuccess);
    CHECK(sip_lqr_plan_create(SIP_LQR_F64, batch, T, n, m, r, &plans[r]) == SIP_LQR_OK);
    const size_t mb = sip_lqr_mats_bytes(plans[r]), vb = sip_lqr_vecs_bytes(plans[r]), gb = sip_lqr_gains_bytes(plans[r]);
    // the default LQRProblem of the reference's tests (lqr_test.cpp:45-75), scaled per rank
    std::vector<double> hm(mb / 8, 0.0), hv(vb / 8, 0.0);
    const size_t stage = (size_t)(n * n + n) + (size_t)(n * n + 2 * n * m + m * m);
    for (int64_t p = 0; p < batch; ++p) {
      double *pm = hm.data() + p * (mb / 8 / batch);
      double *pv = hv.data() + p * (vb / 8 / batch);
      for (int i = 0; i <= T; ++i) {
        double *blk = pm + i * stage;
        for (int d = 0; d < n; ++d)
          blk[d + n * d] = 1.0 + 0.1 * r + 0.01 * p, blk[n * n + d] = 1.0; // Q, delta
        if (i < T) {
          double *e = blk + n * n + n;
          for (int d = 0; d < n; ++d)
            e[d + n * d] = 1.0; // A = I
          for (int k = 0; k < n * m; ++k)
            e[n * n + k] = 1.0; // B = ones
          for (int d = 0; d < m; ++d)
            e[n * n + 2 * n * m + d + m * d] = 1.0; // R = I
        }
        pv[i * (2 * n + m) + n] = 1.0; // c[0] = 1
      }
    }
    CHECK(hipMalloc(&mats[r], mb) == hipSuccess && hipMalloc(&vecs[r], vb) == hipSuccess);
    CHECK(hipMalloc(&sol[r], vb) == hipSuccess && hipMalloc(&gains[r], gb) == hipSuccess);
    CHECK(hipMalloc(&all[r], gb * ndev) == hipSuccess && hipMalloc(&ws[r], sip_lqr_workspace_bytes(plans[r])) == hipSuccess);
    CHECK(hipMalloc((void **)&status[r], sip_lqr_status_bytes(plans[r])) == hipSuccess);
    CHECK(hipMemcpy(mats[r], hm.data(), mb, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(hipMemcpy(vecs[r], hv.data(), vb, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(sip_lqr_factor_solve(plans[r], mats[r], vecs[r], sol[r], gains[r], status[r], ws[r], nullptr) == SIP_LQR_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    host_gains[r].resize(gb / 8);
    CHECK(hipMemcpy(host_gains[r].data(), gains[r], gb, hipMemcpyDeviceToHost) == hipSuccess);
    std::vector<int32_t> st(batch);
    CHECK(hipMemcpy(st.data(), status[r], batch * 4, hipMemcpyDeviceToHost) == hipSuccess);
    for (int32_t s : st)
      CHECK(s == SIP_LQR_SUCCESS);
  }
  CHECK(sip_lqr_group_all_gather_gains(group, plans.data(), gains.data(), all.data(), streams.data()) == SIP_LQR_OK);
  const size_t gl = host_gains[0].size();
  for (int r = 0; r < ndev; ++r) {
    CHECK(hipSetDevice(r) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
    std::vector<double> got(gl * ndev);
    CHECK(hipMemcpy(got.data(), all[r], gl * ndev * 8, hipMemcpyDeviceToHost) == hipSuccess);
    for (int q = 0; q < ndev; ++q)
      CHECK(std::memcmp(got.data() + q * gl, host_gains[q].data(), gl * 8) == 0);
    CHECK(host_gains[r][0] != 0.0);
  }
  std::printf("group of %d device(s): all-gather of %zu gains per rank ok\n", ndev, gl);
  // the same exchange as direct peer copies (rank-major, like the collective above)
  for (int r = 0; r < ndev; ++r)
    CHECK(hipSetDevice(r) == hipSuccess && hipMemset(all[r], 0, gl * ndev * 8) == hipSuccess);
  CHECK(sip_lqr_group_all_gather_gains_p2p(group, plans.data(), gains.data(), all.data(), streams.data()) == SIP_LQR_OK);
  for (int r = 0; r < ndev; ++r) {
    CHECK(hipSetDevice(r) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
    std::vector<double> got(gl * ndev);
    CHECK(hipMemcpy(got.data(), all[r], gl * ndev * 8, hipMemcpyDeviceToHost) == hipSuccess);
    for (int q = 0; q < ndev; ++q)
      CHECK(std::memcmp(got.data() + q * gl, host_gains[q].data(), gl * 8) == 0);
  }
  // chunk-pipelined form: 3 chunks of the 5-problem shard (2 + 2 + 1), chunk-major layout
  const int chunks = 3;
  const size_t per_problem = gl / batch;
  int64_t lo = -1, cnt = -1;
  CHECK(sip_lqr_gains_chunk_range(plans[0], 0, chunks, &lo, &cnt) == SIP_LQR_OK && lo == 0 && cnt == 2);
  CHECK(sip_lqr_gains_chunk_range(plans[0], 2, chunks, &lo, &cnt) == SIP_LQR_OK && lo == 4 && cnt == 1);
  CHECK(sip_lqr_gains_chunk_range(plans[0], 3, chunks, &lo, &cnt) == SIP_LQR_ERR_INVALID_ARGUMENT);
  CHECK(sip_lqr_gains_chunk_offset(plans[0], ndev, 0, 1, chunks) == (size_t)ndev * 2 * per_problem * 8);
  for (int r = 0; r < ndev; ++r)
    CHECK(hipSetDevice(r) == hipSuccess && hipMemset(all[r], 0, gl * ndev * 8) == hipSuccess);
  for (int c = 0; c < chunks; ++c)
    CHECK(sip_lqr_group_all_gather_gains_chunk(group, plans.data(), gains.data(), all.data(), c, chunks,
                                               streams.data()) == SIP_LQR_OK);
  for (int r = 0; r < ndev; ++r) {
    CHECK(hipSetDevice(r) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
    std::vector<double> got(gl * ndev);
    CHECK(hipMemcpy(got.data(), all[r], gl * ndev * 8, hipMemcpyDeviceToHost) == hipSuccess);
    for (int q = 0; q < ndev; ++q)
      for (int c = 0; c < chunks; ++c) {
        CHECK(sip_lqr_gains_chunk_range(plans[q], c, chunks, &lo, &cnt) == SIP_LQR_OK);
        const size_t at = sip_lqr_gains_chunk_offset(plans[q], ndev, q, c, chunks) / 8;
        CHECK(std::memcmp(got.data() + at, host_gains[q].data() + lo * per_problem, cnt * per_problem * 8) == 0);
      }
  }
  std::printf("direct peer copies and %d-chunk exchange ok\n", chunks);
  // the caller's communicator form checks nranks against the communicator (it places the chunk by it)
  // -- no communicator handle is public here, so only the argument checks that come first:
  CHECK(sip_lqr_all_gather_gains_chunk(plans[0], nullptr, ndev, gains[0], all[0], 0, chunks, nullptr) ==
        SIP_LQR_ERR_INVALID_ARGUMENT);
  {
    std::vector<const sip_lqr_plan *> null_plans(ndev, nullptr);
    CHECK(sip_lqr_group_all_gather_gains(group, null_plans.data(), gains.data(), all.data(), streams.data()) ==
          SIP_LQR_ERR_INVALID_ARGUMENT);
    CHECK(sip_lqr_group_all_gather_gains_p2p(group, null_plans.data(), gains.data(), all.data(), streams.data()) ==
          SIP_LQR_ERR_INVALID_ARGUMENT);
  }
  // Peer-copy exchange, chunk form, with completion streams of their own and the send buffer REUSED
  // right behind the call (ncclAllGather's contract): the gains are overwritten on the done stream as
  // soon as the call returns; the gathered data must still be the original.
  {
    std::vector<void *> done(ndev, nullptr);
    for (int r = 0; r < ndev; ++r) {
      hipStream_t st = nullptr;
      CHECK(hipSetDevice(r) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
      done[r] = st;
      CHECK(hipMemset(all[r], 0, gl * ndev * 8) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
    }
    for (int c = 0; c < chunks; ++c)
      CHECK(sip_lqr_group_all_gather_gains_p2p_chunk(group, plans.data(), gains.data(), all.data(), c, chunks,
                                                     streams.data(), done.data()) == SIP_LQR_OK);
    for (int r = 0; r < ndev; ++r) // rewrite the source behind the exchange, on the stream that waited for it
      CHECK(hipSetDevice(r) == hipSuccess && hipMemsetAsync(gains[r], 0xff, gl * 8, (hipStream_t)done[r]) == hipSuccess);
    for (int r = 0; r < ndev; ++r) {
      CHECK(hipSetDevice(r) == hipSuccess && hipStreamSynchronize((hipStream_t)done[r]) == hipSuccess);
      std::vector<double> got(gl * ndev);
      CHECK(hipMemcpy(got.data(), all[r], gl * ndev * 8, hipMemcpyDeviceToHost) == hipSuccess);
      for (int q = 0; q < ndev; ++q)
        for (int c = 0; c < chunks; ++c) {
          CHECK(sip_lqr_gains_chunk_range(plans[q], c, chunks, &lo, &cnt) == SIP_LQR_OK);
          const size_t at = sip_lqr_gains_chunk_offset(plans[q], ndev, q, c, chunks) / 8;
          CHECK(std::memcmp(got.data() + at, host_gains[q].data() + lo * per_problem, cnt * per_problem * 8) == 0);
        }
    }
    for (int r = 0; r < ndev; ++r) { // restore, then the whole-shard form with the source rewritten on streams[r]
      CHECK(hipSetDevice(r) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
      CHECK(hipMemcpy(gains[r], host_gains[r].data(), gl * 8, hipMemcpyHostToDevice) == hipSuccess);
      CHECK(hipMemset(all[r], 0, gl * ndev * 8) == hipSuccess && hipDeviceSynchronize() == hipSuccess);
    }
    CHECK(sip_lqr_group_all_gather_gains_p2p(group, plans.data(), gains.data(), all.data(), done.data()) == SIP_LQR_OK);
    for (int r = 0; r < ndev; ++r)
      CHECK(hipSetDevice(r) == hipSuccess && hipMemsetAsync(gains[r], 0xff, gl * 8, (hipStream_t)done[r]) == hipSuccess);
    for (int r = 0; r < ndev; ++r) {
      CHECK(hipSetDevice(r) == hipSuccess && hipStreamSynchronize((hipStream_t)done[r]) == hipSuccess);
      std::vector<double> got(gl * ndev);
      CHECK(hipMemcpy(got.data(), all[r], gl * ndev * 8, hipMemcpyDeviceToHost) == hipSuccess);
      for (int q = 0; q < ndev; ++q)
        CHECK(std::memcmp(got.data() + q * gl, host_gains[q].data(), gl * 8) == 0);
      CHECK(hipStreamDestroy((hipStream_t)done[r]) == hipSuccess);
    }
    std::printf("peer-copy exchange with completion streams and a reused send buffer ok\n");
  }
  for (int r = 0; r < ndev; ++r)
    sip_lqr_plan_destroy(plans[r]);
  sip_lqr_group_destroy(group);
  std::printf("%d failures\n", g_failures);
  return g_failures == 0 ? 0 : 1;
}
