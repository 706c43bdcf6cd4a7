// tree_lds.hpp -- the general tree engine (tree_generic.hpp) with its per-edge
// working set resident in LDS.
//
// Same algorithm, same step order and the same device routines as
// tree_generic.hpp (one wavefront per problem, lqr.cpp step by step); what
// changes is where the operands live: the blocks of one edge (A, B, M, R, the
// child's F_factor, ...) are copied HBM -> LDS with all loads of a copy in
// flight at once, every product / factorization / substitution then runs on
// LDS operands, and only what the reference keeps (W, K, G_factor per edge; V,
// F_factor, sqrt_delta(_inv), v per node; the solution) is written back.  The
// global-memory version pays one HBM/L2 round trip per scalar of every
// dependent chain; this one pays LDS latency.  Plans whose largest node does
// not fit (lds_scalars() * sizeof(S) > 64 KB) keep the global-memory kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "tree_generic.hpp"

namespace sipamd {
namespace tree {

// scalars of LDS the two kernels need for a plan (max over both)
__host__ __device__ inline long lds_scalars(const int N, const int M) {
  const long factor = 6L * N * N + 4L * N * M + 2L * M * M + 4L * N;
  const long solve = 3L * N * N + 2L * N * M + 1L * M * M + 10L * N + 4L * M;
  return (factor > solve ? factor : solve) + 8;
}

// dst (LDS) <- src (global), len scalars; all loads of a pass issued before the first store.
template <class S>
__device__ __forceinline__ void stage_in(S *dst, const S *__restrict__ src, const int len, const int tid) {
  constexpr int U = 8;
  for (int base = tid; base < len; base += U * TPB) {
    S v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + u * TPB < len)
        v[u] = src[base + u * TPB];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + u * TPB < len)
        dst[base + u * TPB] = v[u];
  }
}

// Several copies at once: the loads of ALL segments are in flight before the first LDS store, so a
// node / an edge costs one HBM round trip instead of one per block.
template <class S, int K>
struct Stager {
  S *dst[K];
  const S *src[K];
  int len[K];
  int count = 0;
  __device__ __forceinline__ void add(S *d, const S *s_, const int l) {
    dst[count] = d, src[count] = s_, len[count] = l;
    ++count;
  }
  __device__ __forceinline__ void run(const int tid) {
    constexpr int U = 2;
    int maxlen = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
      maxlen = (k < count && len[k] > maxlen) ? len[k] : maxlen;
    for (int base = tid; base < maxlen; base += U * TPB) {
      S v[K][U];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k < count && base + u * TPB < len[k])
            v[k][u] = src[k][base + u * TPB];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k < count && base + u * TPB < len[k])
            dst[k][base + u * TPB] = v[k][u];
    }
  }
};

template <class S>
__device__ __forceinline__ void stage_out(S *__restrict__ dst, const S *src, const int len, const int tid) {
  for (int k = tid; k < len; k += TPB)
    dst[k] = src[k];
}

// x <- L^{-1} x for ONE right-hand side, column-oriented so that the lanes share the work; each
// x_i receives its subtractions in the same order (j ascending) as row-oriented substitution.
template <class S>
__device__ inline void solve_lower_vec(const S *L, const int n, S *x, const int tid) {
  for (int j = 0; j < n; ++j) {
    if (tid == 0)
      x[j] = x[j] / L[j + (long)j * n];
    wave_sync();
    const S xj = x[j];
    for (int i = j + 1 + tid; i < n; i += TPB)
      x[i] -= L[i + (long)j * n] * xj;
    wave_sync();
  }
}

// x <- L^{-T} x for one right-hand side.
template <class S>
__device__ inline void solve_lower_t_vec(const S *L, const int n, S *x, const int tid) {
  for (int j = n - 1; j >= 0; --j) {
    if (tid == 0)
      x[j] = x[j] / L[j + (long)j * n];
    wave_sync();
    const S xj = x[j];
    for (int i = tid; i < j; i += TPB)
      x[i] -= L[j + (long)i * n] * xj;
    wave_sync();
  }
}

// lqr.cpp:531-549 on LDS operands
template <class S>
__device__ inline void F_inv_mult_vec(const S *Ffac, const S *rhs, S *res, const S *sd, const S *sdi,
                                      const int n, const int tid) {
  for (int i = tid; i < n; i += TPB)
    res[i] = sdi[i] * rhs[i];
  wave_sync();
  solve_lower_vec(Ffac, n, res, tid);
  solve_lower_t_vec(Ffac, n, res, tid);
  for (int i = tid; i < n; i += TPB)
    res[i] *= sd[i];
  wave_sync();
}

// lqr.cpp:645-731, one problem per workgroup, operands in LDS.
template <class S>
__global__ __launch_bounds__(TPB) void factor_kernel_lds(const Meta mt, const S *in0_all, S *ws_all,
                                                         S *gain_all, int *status, const long batch) {
  extern __shared__ double lds_raw_tree[];
  S *sm = (S *)lds_raw_tree;
  const long b = blockIdx.x;
  if (b >= batch)
    return;
  const int tid = threadIdx.x;
  const long N = mt.max_n, M = mt.max_m;
  S *A = sm, *B = A + N * N, *Mx = B + N * M, *Gf = Mx + N * M, *W = Gf + M * M, *F = W + N * N,
    *H = F + N * N, *K = H + M * N, *V = K + M * N, *Fc = V + N * N, *Fn = Fc + N * N, *dl = Fn + N * N,
    *sd = dl + N, *sdi = sd + N, *sdic = sdi + N;
  const S *in0 = in0_all + b * mt.in0_len;
  S *ws = ws_all + b * mt.ws_len;
  S *gain = gain_all + b * mt.gain_len;
  int result = 0;

  for (int order = 0; order < mt.num_nodes && result == 0; ++order) {
    const int node = mt.postorder[order];
    const int nn = mt.state_dims[node];
    {
      Stager<S, 2> st;
      st.add(V, in0 + mt.oQ[node], nn * nn); // V = Q  (lqr.cpp:658)
      st.add(dl, in0 + mt.od[node], nn);
      st.run(tid);
    }
    wave_sync();

    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1] && result == 0; ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      {
        Stager<S, 6> st;
        st.add(A, in0 + mt.oA[e], nc * nn);
        st.add(B, in0 + mt.oB[e], nc * m);
        st.add(Mx, in0 + mt.oM[e], nn * m);
        st.add(Gf, in0 + mt.oR[e], m * m); // G = R ...  :693
        st.add(Fc, (const S *)(ws + mt.oF[child]), nc * nc);
        st.add(sdic, (const S *)(ws + mt.osdi[child]), nc);
        st.run(tid);
      }
      wave_sync();

      regularized_W(Fc, W, sdic, nc, tid);    // :689
      gemm_tn(m, nc, nc, B, W, S(0), H, tid); // H_child = B^T W  :692
      gemm_nn(m, nc, m, H, B, S(1), Gf, tid); // ... + H_child B  :694
      stage_out(ws + mt.oW[e], W, nc * nc, tid);
      if (chol_lower(Gf, m, tid) >= 0) { // :696-701
        result = 3;
        break;
      }
      stage_out(ws + mt.oG[e], Gf, m * m, tid);
      gemm_nn(nc, nc, nn, W, A, S(0), F, tid); // F = W A  :703
      for (int idx = tid; idx < m * nn; idx += TPB) {
        const int row = idx % m, col = idx / m;
        H[idx] = Mx[col + (long)row * nn]; // H_parent = M^T  :704
      }
      wave_sync();
      gemm_tn(m, nc, nn, B, F, S(1), H, tid); // += B^T F  :705
      for (int idx = tid; idx < m * nn; idx += TPB)
        K[idx] = H[idx]; // :707
      wave_sync();
      solve_lower(Gf, m, K, nn, tid);   // :708
      solve_lower_t(Gf, m, K, nn, tid); // :710
      for (int idx = tid; idx < m * nn; idx += TPB)
        K[idx] = -K[idx]; // :713
      wave_sync();
      stage_out(gain + mt.oK[e], K, m * nn, tid);
      gemm_tn(nn, nc, nn, A, F, S(1), V, tid); // V += A^T F  :715
      gemm_tn(nn, m, nn, K, H, S(0), F, tid);  // F_parent = K^T H  :718
      for (int idx = tid; idx < nn * nn; idx += TPB)
        V[idx] += F[idx]; // :719
      wave_sync();
    }
    if (result != 0)
      break;
    stage_out(ws + mt.oV[node], V, nn * nn, tid);
    result = factor_F(dl, V, Fn, sd, sdi, nn, tid); // :722-727
    if (result == 0) {
      stage_out(ws + mt.oF[node], Fn, nn * nn, tid);
      stage_out(ws + mt.osd[node], sd, nn, tid);
      stage_out(ws + mt.osdi[node], sdi, nn, tid);
    }
    wave_sync(); // the parent re-reads F_factor / sqrt_delta_inv of this node from memory
  }
  if (tid == 0)
    status[b] = result;
}

// lqr.cpp:735-871, one problem per workgroup, operands in LDS.
template <class S>
__global__ __launch_bounds__(TPB) void solve_kernel_lds(const Meta mt, const S *in0_all, const S *in1_all,
                                                        S *ws_all, S *gain_all, S *out_all,
                                                        const int *status, const long batch) {
  extern __shared__ double lds_raw_tree[];
  S *sm = (S *)lds_raw_tree;
  const long b = blockIdx.x;
  if (b >= batch || status[b] != 0)
    return;
  const int tid = threadIdx.x;
  const long N = mt.max_n, M = mt.max_m;
  // matrices: A | W (or V) | F_factor | B | K | G_factor ; then the vectors
  S *A = sm, *W = A + N * N, *Ff = W + N * N, *B = Ff + N * N, *K = B + N * M, *Gf = K + N * M,
    *f = Gf + M * M, *g = f + N, *v = g + N, *vc = v + N, *cc = vc + N, *dc = cc + N, *sdv = dc + N,
    *sdiv = sdv + N, *xn = sdiv + N, *xc = xn + N, *h = xc + N, *k = h + M, *r = k + M, *u = r + M;
  const S *in0 = in0_all + b * mt.in0_len;
  const S *in1 = in1_all + b * mt.in1_len;
  S *ws = ws_all + b * mt.ws_len;
  S *gain = gain_all + b * mt.gain_len;
  S *out = out_all + b * mt.out_len;

  for (int order = 0; order < mt.num_nodes; ++order) { // backward affine sweep  :738-796
    const int node = mt.postorder[order];
    const int nn = mt.state_dims[node];
    stage_in(v, in1 + mt.oq[node], nn, tid); // v = q
    wave_sync();
    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1]; ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      {
        Stager<S, 9> st;
        st.add(A, in0 + mt.oA[e], nc * nn);
        st.add(B, in0 + mt.oB[e], nc * m);
        st.add(W, (const S *)(ws + mt.oW[e]), nc * nc);
        st.add(Gf, (const S *)(ws + mt.oG[e]), m * m);
        st.add(K, (const S *)(gain + mt.oK[e]), m * nn);
        st.add(vc, (const S *)(ws + mt.ov[child]), nc);
        st.add(cc, in1 + mt.oc[child], nc);
        st.add(dc, in0 + mt.od[child], nc);
        st.add(r, in1 + mt.orr[e], m);
        st.run(tid);
      }
      wave_sync();
      for (int i = tid; i < nc; i += TPB)
        f[i] = dc[i] * vc[i] - cc[i]; // :778-779
      wave_sync();
      for (int i = tid; i < nc; i += TPB) { // g = v_c - W f  :780-781
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += W[i + (long)j * nc] * f[j];
        g[i] = vc[i] - s;
      }
      wave_sync();
      for (int i = tid; i < m; i += TPB) { // h = r + B^T g  :783-784
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += B[j + (long)i * nc] * g[j];
        h[i] = r[i] + s;
        k[i] = h[i]; // :785
      }
      wave_sync();
      solve_lower_vec(Gf, m, k, tid);
      solve_lower_t_vec(Gf, m, k, tid);
      for (int i = tid; i < m; i += TPB) {
        k[i] = -k[i]; // :791
        (gain + mt.ok[e])[i] = k[i];
      }
      for (int i = tid; i < nn; i += TPB) { // v += A^T g + K^T h  :793-794
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += A[j + (long)i * nc] * g[j];
        for (int j = 0; j < m; ++j)
          s += K[j + (long)i * m] * h[j];
        v[i] += s;
      }
      wave_sync();
    }
    stage_out(ws + mt.ov[node], v, nn, tid);
    wave_sync(); // the parent re-reads v of this node from memory
  }

  { // root  :798-819
    const int root = mt.preorder[0];
    const int n = mt.state_dims[root];
    {
      Stager<S, 7> st;
      st.add(W, (const S *)(ws + mt.oV[root]), n * n);
      st.add(Ff, (const S *)(ws + mt.oF[root]), n * n);
      st.add(sdv, (const S *)(ws + mt.osd[root]), n);
      st.add(sdiv, (const S *)(ws + mt.osdi[root]), n);
      st.add(vc, (const S *)(ws + mt.ov[root]), n);
      st.add(cc, in1 + mt.oc[root], n);
      st.add(dc, in0 + mt.od[root], n);
      st.run(tid);
    }
    wave_sync();
    for (int i = tid; i < n; i += TPB)
      f[i] = dc[i] * vc[i] - cc[i];
    wave_sync();
    F_inv_mult_vec(Ff, f, xc, sdv, sdiv, n, tid);
    for (int i = tid; i < n; i += TPB)
      xc[i] = -xc[i];
    wave_sync();
    for (int i = tid; i < n; i += TPB) {
      S s = S(0);
      for (int j = 0; j < n; ++j)
        s += W[i + (long)j * n] * xc[j];
      (out + mt.oy[root])[i] = vc[i] + s;
      (out + mt.ox[root])[i] = xc[i];
    }
    wave_sync();
  }

  for (int order = 0; order < mt.num_nodes; ++order) { // rollout  :821-870
    const int node = mt.preorder[order];
    const int nn = mt.state_dims[node];
    if (mt.child_offsets[node] == mt.child_offsets[node + 1])
      continue;
    stage_in(xn, (const S *)(out + mt.ox[node]), nn, tid);
    wave_sync();
    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1]; ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      {
        Stager<S, 11> st;
        st.add(A, in0 + mt.oA[e], nc * nn);
        st.add(B, in0 + mt.oB[e], nc * m);
        st.add(K, (const S *)(gain + mt.oK[e]), m * nn);
        st.add(k, (const S *)(gain + mt.ok[e]), m);
        st.add(W, (const S *)(ws + mt.oV[child]), nc * nc); // V of the child
        st.add(Ff, (const S *)(ws + mt.oF[child]), nc * nc);
        st.add(sdv, (const S *)(ws + mt.osd[child]), nc);
        st.add(sdiv, (const S *)(ws + mt.osdi[child]), nc);
        st.add(vc, (const S *)(ws + mt.ov[child]), nc);
        st.add(cc, in1 + mt.oc[child], nc);
        st.add(dc, in0 + mt.od[child], nc);
        st.run(tid);
      }
      wave_sync();
      for (int i = tid; i < m; i += TPB) { // u = k + K x  :856-857
        S s = S(0);
        for (int j = 0; j < nn; ++j)
          s += K[i + (long)j * m] * xn[j];
        u[i] = k[i] + s;
        (out + mt.ou[e])[i] = u[i];
      }
      wave_sync();
      for (int i = tid; i < nc; i += TPB) { // :859-862
        S s = cc[i] - dc[i] * vc[i];
        for (int j = 0; j < nn; ++j)
          s += A[i + (long)j * nc] * xn[j];
        for (int j = 0; j < m; ++j)
          s += B[i + (long)j * nc] * u[j];
        f[i] = s;
      }
      wave_sync();
      F_inv_mult_vec(Ff, f, xc, sdv, sdiv, nc, tid); // :863-865
      for (int i = tid; i < nc; i += TPB) {          // y = v + V x  :867-868
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += W[i + (long)j * nc] * xc[j];
        (out + mt.oy[child])[i] = vc[i] + s;
        (out + mt.ox[child])[i] = xc[i];
      }
      wave_sync(); // x of the child is re-read from memory when it becomes a parent
    }
  }
}

} // namespace tree
} // namespace sipamd
