// generic_plan.hpp -- host side of the general (tree_generic.hpp) engine:
// offset tables for a given arena layout, their upload, kernel launches.
// Shared by the tree C ABI (tree-native arenas) and by the chain C ABI (packed
// chain layout: shapes / dtypes without a dedicated kernel, and the split
// factor / solve entry points).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include <cstdlib>
#include <cstring>

#include "stream_fill.hpp"
#include "tree_generic.hpp"
#include "tree_lds.hpp"

namespace sipamd {

struct GenericPlan {
  int E = 0, N = 0, root = 0, max_n = 0, max_m = 0;
  std::vector<int> state_dims, control_dims, parents, children;
  std::vector<int> child_offsets, child_edges, preorder, postorder;
  // per-node / per-edge offsets, see tree::Meta
  std::vector<long> oQ, od, oA, oB, oM, oR, oq, oc, orr, ox, oy, ou, oK, ok, oW, oG, oV, oF, osd,
      osdi, ov;
  long scratch_ws = 0, in0_len = 0, in1_len = 0, out_len = 0, gain_len = 0, ws_len = 0;
  void *d_ints = nullptr, *d_longs = nullptr;
  tree::Meta meta{};

  ~GenericPlan() {
    if (d_ints)
      (void)hipFree(d_ints);
    if (d_longs)
      (void)hipFree(d_longs);
  }

  void set_shape(int num_edges, int root_node, const int *sd, const int *cd) {
    E = num_edges, N = num_edges + 1, root = root_node;
    state_dims.assign(sd, sd + N);
    control_dims.assign(cd, cd + E);
    max_n = N ? *std::max_element(state_dims.begin(), state_dims.end()) : 0;
    max_m = E ? *std::max_element(control_dims.begin(), control_dims.end()) : 0;
    for (auto *v : {&oQ, &od, &oq, &oc, &ox, &oy, &oV, &oF, &osd, &osdi, &ov})
      v->assign(N, 0);
    for (auto *v : {&oA, &oB, &oM, &oR, &orr, &ou, &oK, &ok, &oW, &oG})
      v->assign(E, 0);
  }
  long np(int e) const { return state_dims[parents[e]]; }
  long nc(int e) const { return state_dims[children[e]]; }

  // Work arena shared by every layout: per edge W (max_n^2) | G_factor (m^2),
  // per node V | F_factor | sqrt_delta | sqrt_delta_inv | v, then the scratch.
  // `start`: first free scalar of the work arena.
  void layout_work(long start) {
    long ws = start;
    for (int e = 0; e < E; ++e) {
      const long m = control_dims[e];
      oW[e] = ws, ws += (long)max_n * max_n;
      oG[e] = ws, ws += m * m;
    }
    for (int i = 0; i < N; ++i) {
      const long n = state_dims[i];
      oV[i] = ws, ws += n * n;
      oF[i] = ws, ws += n * n;
      osd[i] = ws, ws += n;
      osdi[i] = ws, ws += n;
      ov[i] = ws, ws += n;
    }
    scratch_ws = ws;
    ws += (long)max_m * max_n + (long)max_n * max_n + 2L * max_n + max_m;
    ws_len = ws;
  }

  // Tree-native arenas (include/sip_lqr_amd.h, second half).  One input arena
  // (in1 == in0), gains inside the work arena with the reference's
  // LQR::Workspace field order per edge: W | K | G_factor | k.
  void layout_tree_native() {
    long in = 0, o = 0, ws = 0;
    for (int i = 0; i < N; ++i) {
      const long n = state_dims[i];
      oQ[i] = in, in += n * n;
      oq[i] = in, in += n;
      oc[i] = in, in += n;
      od[i] = in, in += n;
      ox[i] = o, o += n;
      oy[i] = o, o += n;
    }
    for (int e = 0; e < E; ++e) {
      const long m = control_dims[e];
      oA[e] = in, in += nc(e) * np(e);
      oB[e] = in, in += nc(e) * m;
      oM[e] = in, in += np(e) * m;
      oR[e] = in, in += m * m;
      orr[e] = in, in += m;
      ou[e] = o, o += m;
    }
    in0_len = in1_len = in, out_len = o;
    for (int e = 0; e < E; ++e) {
      const long m = control_dims[e];
      oW[e] = ws, ws += (long)max_n * max_n;
      oK[e] = ws, ws += m * np(e);
      oG[e] = ws, ws += m * m;
      ok[e] = ws, ws += m;
    }
    for (int i = 0; i < N; ++i) {
      const long n = state_dims[i];
      oV[i] = ws, ws += n * n;
      oF[i] = ws, ws += n * n;
      osd[i] = ws, ws += n;
      osdi[i] = ws, ws += n;
      ov[i] = ws, ws += n;
    }
    scratch_ws = ws;
    ws += (long)max_m * max_n + (long)max_n * max_n + 2L * max_n + max_m;
    ws_len = gain_len = ws; // the gain arena IS the work arena
  }

  // Packed chain layout (include/sip_lqr_amd.h, first half): uniform n, m,
  // chain topology; in0 = mats, in1 = vecs, out = sol, gain = gains.
  void layout_chain(int n, int m, int T) {
    const long node = (long)n * n + n, edge = (long)n * n + 2L * n * m + (long)m * m;
    const long vstg = 2L * n + m, gstg = (long)m * n + m;
    for (int i = 0; i <= T; ++i) {
      const long mb = i * (node + edge), vb = i * vstg;
      oQ[i] = mb, od[i] = mb + (long)n * n;
      oq[i] = vb, oc[i] = vb + n;
      ox[i] = vb, oy[i] = vb + n;
      if (i < T) {
        const long eb = mb + node;
        oA[i] = eb, oB[i] = eb + (long)n * n, oM[i] = oB[i] + (long)n * m, oR[i] = oM[i] + (long)n * m;
        orr[i] = vb + 2L * n;
        ou[i] = vb + 2L * n;
        oK[i] = i * gstg, ok[i] = i * gstg + (long)m * n;
      }
    }
    in0_len = (T + 1) * node + T * edge;
    in1_len = out_len = (T + 1) * 2L * n + (long)T * m;
    gain_len = T * gstg;
    layout_work(0);
  }

  hipError_t upload(int device) {
    std::vector<int> ints;
    auto pi = [&](const std::vector<int> &v) {
      const size_t at = ints.size();
      ints.insert(ints.end(), v.begin(), v.end());
      return at;
    };
    const size_t a_sd = pi(state_dims), a_cd = pi(control_dims), a_pa = pi(parents), a_ch = pi(children),
                 a_co = pi(child_offsets), a_ce = pi(child_edges), a_pre = pi(preorder), a_post = pi(postorder);
    std::vector<long> longs;
    std::vector<size_t> at;
    const std::vector<long> *tabs[] = {&oQ, &od, &oA, &oB, &oM, &oR, &oq, &oc, &orr, &ox, &oy,
                                       &ou, &oK, &ok, &oW, &oG, &oV, &oF, &osd, &osdi, &ov};
    for (const auto *t : tabs) {
      at.push_back(longs.size());
      longs.insert(longs.end(), t->begin(), t->end());
    }
    DeviceGuard on_device(device); // the caller's current device is restored on return
    hipError_t e = on_device.err;
    if (e != hipSuccess) return e;
    if ((e = hipMalloc(&d_ints, std::max<size_t>(1, ints.size()) * sizeof(int))) != hipSuccess) return e;
    if ((e = hipMalloc(&d_longs, std::max<size_t>(1, longs.size()) * sizeof(long))) != hipSuccess) return e;
    if ((e = hipMemcpy(d_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess)
      return e;
    if ((e = hipMemcpy(d_longs, longs.data(), longs.size() * sizeof(long), hipMemcpyHostToDevice)) != hipSuccess)
      return e;
    const int *di = (const int *)d_ints;
    const long *dl = (const long *)d_longs;
    tree::Meta &m = meta;
    m.num_edges = E, m.num_nodes = N, m.root = root, m.max_n = max_n, m.max_m = max_m;
    m.state_dims = di + a_sd, m.control_dims = di + a_cd, m.edge_parents = di + a_pa;
    m.edge_children = di + a_ch, m.child_offsets = di + a_co, m.child_edges = di + a_ce;
    m.preorder = di + a_pre, m.postorder = di + a_post;
    const long **dst[] = {&m.oQ, &m.od, &m.oA, &m.oB, &m.oM, &m.oR, &m.oq, &m.oc, &m.orr, &m.ox, &m.oy,
                          &m.ou, &m.oK, &m.ok, &m.oW, &m.oG, &m.oV, &m.oF, &m.osd, &m.osdi, &m.ov};
    for (size_t i = 0; i < at.size(); ++i)
      *dst[i] = dl + at[i];
    m.scratch_ws = scratch_ws;
    m.in0_len = in0_len, m.in1_len = in1_len, m.out_len = out_len, m.gain_len = gain_len, m.ws_len = ws_len;
    return hipSuccess;
  }

  // LDS-resident kernels (tree_lds.hpp) when the largest node fits in 64 KB; SIP_LQR_TREE=global
  // keeps the global-memory kernels (tests, A/B).
  template <class S>
  size_t lds_bytes() const {
    const char *v = std::getenv("SIP_LQR_TREE");
    if (v != nullptr && std::strcmp(v, "global") == 0)
      return 0;
    const size_t need = (size_t)tree::lds_scalars(max_n, max_m) * sizeof(S);
    return need <= 64 * 1024 ? need : 0;
  }

  template <class S>
  hipError_t launch_factor(long batch, const void *in0, void *ws, void *gain, int32_t *status,
                           hipStream_t stream) const {
    if (const size_t lds = lds_bytes<S>())
      hipLaunchKernelGGL((tree::factor_kernel_lds<S>), dim3((unsigned)batch), dim3(tree::TPB), lds, stream, meta,
                         (const S *)in0, (S *)ws, (S *)gain, (int *)status, batch);
    else
      hipLaunchKernelGGL((tree::factor_kernel<S>), dim3((unsigned)batch), dim3(tree::TPB), 0, stream, meta,
                         (const S *)in0, (S *)ws, (S *)gain, (int *)status, batch);
    return hipGetLastError();
  }
  template <class S>
  hipError_t launch_solve(long batch, const void *in0, const void *in1, void *ws, void *gain, void *out,
                          const int32_t *status, hipStream_t stream) const {
    if (const size_t lds = lds_bytes<S>())
      hipLaunchKernelGGL((tree::solve_kernel_lds<S>), dim3((unsigned)batch), dim3(tree::TPB), lds, stream, meta,
                         (const S *)in0, (const S *)in1, (S *)ws, (S *)gain, (S *)out, (const int *)status,
                         batch);
    else
      hipLaunchKernelGGL((tree::solve_kernel<S>), dim3((unsigned)batch), dim3(tree::TPB), 0, stream, meta,
                         (const S *)in0, (const S *)in1, (S *)ws, (S *)gain, (S *)out, (const int *)status,
                         batch);
    return hipGetLastError();
  }
};

} // namespace sipamd
