// Development aid: the n = 32 matrix-core chain kernel (chain_mt16.hpp) built with segment stamps, on
// synthetic well-posed data (the timing does not depend on the values), batch 4096, T = 100.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DSIP_MT16_STAMPS \
//         tools/dev/mt16_stamps.hip -o tools/dev/mt16_stamps && tools/dev/mt16_stamps [f64]
#include "../../sip_optimal_control_amd/csrc/chain_mt16.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

template <typename S> int run(long batch, int T) {
  using namespace sipamd::mt16;
  constexpr int M = 8;
  using LY = Layout<S, M>;
  const long mats_len = (long)(T + 1) * LY::NODE + (long)T * LY::EDGE, vecs_len = (long)(T + 1) * LY::VNODE + (long)T * LY::VEDGE;
  std::vector<S> hm(mats_len), hv(vecs_len);
  std::mt19937 rng(1);
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> ud;
  for (int i = 0; i <= T; ++i) {
    S *node = hm.data() + (long)i * (LY::NODE + LY::EDGE);
    for (int c = 0; c < N; ++c)
      for (int r = 0; r < N; ++r)
        node[c * N + r] = (S)((r == c ? 3.0 : 0.0) + 0.02 * std::cos(0.37 * (r + 1) * (c + 1) + i)); // symmetric
    for (int r = 0; r < N; ++r)
      node[N * N + r] = (S)(1e-3 + 0.1 * ud(rng));
    if (i < T) {
      S *e = node + LY::NODE;
      for (int k = 0; k < N * N; ++k)
        e[k] = (S)((k % (N + 1) == 0 ? 1.0 : 0.0) + 0.05 * nd(rng));
      for (int k = 0; k < 2 * N * M; ++k)
        e[N * N + k] = (S)(0.1 * nd(rng));
      for (int c = 0; c < M; ++c)
        for (int r = 0; r < M; ++r)
          e[N * N + 2 * N * M + c * M + r] = (S)(r == c ? 2.0 : 0.1 / (1 + r + c));
    }
  }
  for (auto &v : hv)
    v = (S)nd(rng);
  S *mats, *vecs, *sol, *gains, *ws;
  int *status;
  unsigned long long *stamps;
  hipMalloc(&mats, batch * mats_len * sizeof(S)), hipMalloc(&vecs, batch * vecs_len * sizeof(S));
  hipMalloc(&sol, batch * vecs_len * sizeof(S)), hipMalloc(&gains, batch * (long)T * LY::GAIN * sizeof(S));
  hipMalloc(&ws, batch * (long)(T + 1) * LY::WSN * sizeof(S)), hipMalloc(&status, batch * 4);
  hipMalloc(&stamps, batch * 20 * 8);
  for (long p = 0; p < batch; ++p) {
    hipMemcpy(mats + p * mats_len, hm.data(), mats_len * sizeof(S), hipMemcpyHostToDevice);
    hipMemcpy(vecs + p * vecs_len, hv.data(), vecs_len * sizeof(S), hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain_factor_solve_mt16<S, M>), dim3((unsigned)batch), dim3(64), 0, 0, mats, vecs, sol, gains, ws,
                       status, batch, T, stamps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<int> st(batch);
  hipMemcpy(st.data(), status, batch * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int s : st)
    bad += s != 0;
  std::vector<unsigned long long> h(batch * 20);
  hipMemcpy(h.data(), stamps, batch * 20 * 8, hipMemcpyDeviceToHost);
  auto med = [&](auto f) {
    std::vector<double> v(batch);
    for (long p = 0; p < batch; ++p)
      v[p] = f(h.data() + p * 20);
    std::nth_element(v.begin(), v.begin() + batch / 2, v.end());
    return v[batch / 2];
  };
  std::printf("%s batch %ld T %d: %.3f ms, %ld problems with status != 0\n", sizeof(S) == 4 ? "f32" : "f64", batch, T, ms, bad);
  std::printf("  cycles per wave: backward %.0f (per stage %.0f), root + rollout %.0f (per stage %.0f)\n",
              med([](const unsigned long long *o) { return (double)(o[1] - o[0]); }),
              med([](const unsigned long long *o) { return (double)(o[1] - o[0]); }) / T,
              med([](const unsigned long long *o) { return (double)(o[2] - o[1]); }),
              med([](const unsigned long long *o) { return (double)(o[2] - o[1]); }) / T);
  {
    unsigned long long t0 = ~0ull;
    for (long p = 0; p < batch; ++p)
      t0 = std::min(t0, h[p * 20]);
    std::vector<double> st(batch), du(batch), en(batch);
    for (long p = 0; p < batch; ++p)
      st[p] = (double)(h[p * 20] - t0), du[p] = (double)(h[p * 20 + 2] - h[p * 20]), en[p] = (double)(h[p * 20 + 2] - t0);
    std::sort(st.begin(), st.end()), std::sort(du.begin(), du.end()), std::sort(en.begin(), en.end());
    auto q = [&](std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    std::printf("  start  (cycles after the first wave): 10%% %.0f  50%% %.0f  75%% %.0f  90%% %.0f  max %.0f\n", q(st, .1), q(st, .5), q(st, .75), q(st, .9), q(st, 1));
    std::printf("  lifetime of a wave:                  min %.0f  10%% %.0f  50%% %.0f  90%% %.0f  max %.0f\n", q(du, 0), q(du, .1), q(du, .5), q(du, .9), q(du, 1));
    std::printf("  end    (cycles after the first wave): 10%% %.0f  50%% %.0f  90%% %.0f  max %.0f\n", q(en, .1), q(en, .5), q(en, .9), q(en, 1));
  }
  const char *names[] = {"loads, g = v + W t", "F = W A, Z = W B, G, H (MFMA)", "h = r + B^T g", "scale + sweep G", "K, k, gains, v",
                         "V = Q + A^T F + K^T H", "-", "node: delta, t", "mirror V, F = I + sd V sd", "sweep F", "W, spill"};
  for (int k = 0; k < 11; ++k)
    std::printf("    %-34s %8.0f per stage\n", names[k], med([k](const unsigned long long *o) { return (double)o[3 + k]; }) / T);
  return 0;
}

int main(int argc, char **argv) {
  const bool f64 = argc > 1 && std::strcmp(argv[1], "f64") == 0;
  const long batch = argc > 2 ? std::atol(argv[2]) : 4096;
  return f64 ? run<double>(batch, 100) : run<float>(batch, 100);
}
