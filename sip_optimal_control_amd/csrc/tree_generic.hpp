// tree_generic.hpp -- reference-shaped regularized tree-LQR on the GPU: any tree
// topology, per-node state / per-edge control dimensions (0 allowed), fp64 or fp32.
//
// One wavefront (one 64-thread workgroup) per problem instance; the batch is
// `batch` instances of ONE topology / dimension table (the plan), each with its
// own data.  The factor state lives in a device workspace with exactly the
// fields of the reference's LQR::Workspace (lqr.hpp:110-127: W, K, G_factor, k
// per edge; V, F_factor, sqrt_delta, sqrt_delta_inv, v per node; scratch H, F,
// g, h, f), so that consumers of those fields (helpers.cpp:521-665) can be fed
// from it.  This is the general path behind the drop-in `LQR` adapter
// (include/sip_optimal_control_amd/lqr_dropin.hpp); the uniform-chain batch
// kernel (chain_qw16.hpp) is the throughput path.
//
// The arithmetic follows lqr.cpp step by step: Cholesky is the unblocked
// left-looking algorithm of Eigen::LLT (pivot x = a_kk - |L_k,0:k|^2, failure
// iff x <= 0, IEEE sqrt and divide), triangular solves are plain substitution.
#pragma once
#include <hip/hip_runtime.h>

namespace sipamd {
namespace tree {

// Device-side description of the plan (arrays in device memory, shared by all
// problems of the batch).  Every block is addressed by an explicit offset (in
// scalars, from the start of one problem inside its arena), so the same
// kernels serve the tree-native arenas (include/sip_lqr_amd.h, second half)
// and the packed chain layout (first half).  Arenas:
//   in0  : Q, delta (per node); A, B, M, R (per edge)     [chain: mats]
//   in1  : q, c (per node); r (per edge)                   [chain: vecs]
//   out  : x, y (per node); u (per edge)                   [chain: sol]
//   gain : K, k (per edge)                                  [chain: gains]
//   work : W, G_factor (per edge); V, F_factor, sqrt_delta, sqrt_delta_inv,
//          v (per node); scratch H | F | g | h | f
struct Meta {
  int num_edges, num_nodes, root, max_n, max_m;
  const int *state_dims, *control_dims;
  const int *edge_parents, *edge_children;
  const int *child_offsets, *child_edges, *preorder, *postorder;
  const long *oQ, *od, *oA, *oB, *oM, *oR; // in0
  const long *oq, *oc, *orr;               // in1
  const long *ox, *oy, *ou;                // out
  const long *oK, *ok;                     // gain
  const long *oW, *oG;                     // work, per edge
  const long *oV, *oF, *osd, *osdi, *ov;   // work, per node
  long scratch_ws;                         // H | F | g | h | f
  long in0_len, in1_len, out_len, gain_len, ws_len;
};

constexpr int TPB = 64;

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

__device__ __forceinline__ double sqrt_s(double x) { return sqrt(x); }
__device__ __forceinline__ float sqrt_s(float x) { return sqrtf(x); }

// In-place lower Cholesky (Eigen LLT unblocked).  Uniform return: failing
// pivot index or -1.
template <class S>
__device__ inline int chol_lower(S *a, const int n, const int tid) {
  for (int k = 0; k < n; ++k) {
    S x = a[k + (long)k * n];
    for (int j = 0; j < k; ++j)
      x -= a[k + (long)j * n] * a[k + (long)j * n];
    if (x <= S(0))
      return k;
    const S d = sqrt_s(x);
    for (int i = k + 1 + tid; i < n; i += TPB) {
      S s = a[i + (long)k * n];
      for (int j = 0; j < k; ++j)
        s -= a[i + (long)j * n] * a[k + (long)j * n];
      a[i + (long)k * n] = s / d;
    }
    wave_sync(); // everyone has read the old a_kk before it becomes L_kk
    if (tid == 0)
      a[k + (long)k * n] = d;
    wave_sync();
  }
  return -1;
}

// X <- L^{-1} X, one right-hand-side column per thread.
template <class S>
__device__ inline void solve_lower(const S *L, const int n, S *X, const int nrhs,
                                   const int tid) {
  for (int col = tid; col < nrhs; col += TPB) {
    S *x = X + (long)col * n;
    for (int i = 0; i < n; ++i) {
      S s = x[i];
      for (int j = 0; j < i; ++j)
        s -= L[i + (long)j * n] * x[j];
      x[i] = s / L[i + (long)i * n];
    }
  }
  wave_sync();
}

// X <- L^{-T} X.
template <class S>
__device__ inline void solve_lower_t(const S *L, const int n, S *X,
                                     const int nrhs, const int tid) {
  for (int col = tid; col < nrhs; col += TPB) {
    S *x = X + (long)col * n;
    for (int i = n - 1; i >= 0; --i) {
      S s = x[i];
      for (int j = i + 1; j < n; ++j)
        s -= L[j + (long)i * n] * x[j];
      x[i] = s / L[i + (long)i * n];
    }
  }
  wave_sync();
}

// C (p x r) = beta C + A^T B, A (q x p), B (q x r).
template <class S>
__device__ inline void gemm_tn(const int p, const int q, const int r, const S *A,
                               const S *B, const S beta, S *C, const int tid) {
  for (int idx = tid; idx < p * r; idx += TPB) {
    const int i = idx % p, j = idx / p;
    S s = S(0);
    for (int l = 0; l < q; ++l)
      s += A[l + (long)i * q] * B[l + (long)j * q];
    C[idx] = (beta == S(0) ? S(0) : beta * C[idx]) + s;
  }
  wave_sync();
}

// C (p x r) = beta C + A B, A (p x q), B (q x r).
template <class S>
__device__ inline void gemm_nn(const int p, const int q, const int r, const S *A,
                               const S *B, const S beta, S *C, const int tid) {
  for (int idx = tid; idx < p * r; idx += TPB) {
    const int i = idx % p, j = idx / p;
    S s = S(0);
    for (int l = 0; l < q; ++l)
      s += A[i + (long)l * p] * B[l + (long)j * q];
    C[idx] = (beta == S(0) ? S(0) : beta * C[idx]) + s;
  }
  wave_sync();
}

// lqr.cpp:487-509.  Returns 0 / INVALID_DELTA(1) / F_FACTORIZATION_FAILURE(2).
template <class S>
__device__ inline int factor_F(const S *delta, const S *V, S *F, S *sd, S *sdi,
                               const int n, const int tid) {
  for (int i = 0; i < n; ++i) // uniform scan, as the reference's early return
    if (delta[i] <= S(0))
      return 1;
  for (int i = tid; i < n; i += TPB) {
    sd[i] = sqrt_s(delta[i]);
    sdi[i] = S(1) / sd[i];
  }
  wave_sync();
  for (int idx = tid; idx < n * n; idx += TPB) {
    const int row = idx % n, col = idx / n;
    F[idx] = sd[row] * V[idx] * sd[col] + (row == col ? S(1) : S(0));
  }
  wave_sync();
  return chol_lower(F, n, tid) < 0 ? 0 : 2;
}

// lqr.cpp:511-529.
template <class S>
__device__ inline void regularized_W(const S *Ffac, S *W, const S *sdi,
                                     const int n, const int tid) {
  for (int idx = tid; idx < n * n; idx += TPB)
    W[idx] = (idx % n == idx / n) ? S(1) : S(0);
  wave_sync();
  solve_lower(Ffac, n, W, n, tid);
  solve_lower_t(Ffac, n, W, n, tid);
  for (int idx = tid; idx < n * n; idx += TPB) {
    const int row = idx % n, col = idx / n;
    S w = -W[idx];
    if (row == col)
      w += S(1);
    W[idx] = w * (sdi[row] * sdi[col]);
  }
  wave_sync();
}

// lqr.cpp:531-549.
template <class S>
__device__ inline void F_inv_mult(const S *Ffac, const S *rhs, S *res,
                                  const S *sd, const S *sdi, const int n,
                                  const int tid) {
  for (int i = tid; i < n; i += TPB)
    res[i] = sdi[i] * rhs[i];
  wave_sync();
  solve_lower(Ffac, n, res, 1, tid);
  solve_lower_t(Ffac, n, res, 1, tid);
  for (int i = tid; i < n; i += TPB)
    res[i] *= sd[i];
  wave_sync();
}

// lqr.cpp:645-731, one problem per workgroup.
template <class S>
__global__ __launch_bounds__(TPB) void factor_kernel(const Meta mt,
                                                     const S *in0_all,
                                                     S *ws_all, S *gain_all,
                                                     int *status,
                                                     const long batch) {
  const long b = blockIdx.x;
  if (b >= batch)
    return;
  const int tid = threadIdx.x;
  const S *in0 = in0_all + b * mt.in0_len;
  S *ws = ws_all + b * mt.ws_len;
  S *gain = gain_all + b * mt.gain_len;
  S *H = ws + mt.scratch_ws;
  S *F = H + (long)mt.max_m * mt.max_n;
  int result = 0;

  for (int order = 0; order < mt.num_nodes && result == 0; ++order) {
    const int node = mt.postorder[order];
    const int nn = mt.state_dims[node];
    const S *Q = in0 + mt.oQ[node];
    S *V = ws + mt.oV[node];
    for (int idx = tid; idx < nn * nn; idx += TPB)
      V[idx] = Q[idx]; // V = Q  (lqr.cpp:658)
    wave_sync();

    for (int ci = mt.child_offsets[node];
         ci < mt.child_offsets[node + 1] && result == 0; ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      const S *A = in0 + mt.oA[e], *B = in0 + mt.oB[e], *M = in0 + mt.oM[e],
              *R = in0 + mt.oR[e];
      S *W = ws + mt.oW[e], *Gf = ws + mt.oG[e], *K = gain + mt.oK[e];
      const S *Fc = ws + mt.oF[child], *sdic = ws + mt.osdi[child];

      regularized_W(Fc, W, sdic, nc, tid);         // :689
      gemm_tn(m, nc, nc, B, W, S(0), H, tid);      // H_child = B^T W  :692
      for (int idx = tid; idx < m * m; idx += TPB)
        Gf[idx] = R[idx];                          // :693
      wave_sync();
      gemm_nn(m, nc, m, H, B, S(1), Gf, tid);      // :694
      if (chol_lower(Gf, m, tid) >= 0) {           // :696-701
        result = 3;
        break;
      }
      gemm_nn(nc, nc, nn, W, A, S(0), F, tid);     // F = W A  :703
      for (int idx = tid; idx < m * nn; idx += TPB) {
        const int row = idx % m, col = idx / m;
        H[idx] = M[col + (long)row * nn];          // H_parent = M^T  :704
      }
      wave_sync();
      gemm_tn(m, nc, nn, B, F, S(1), H, tid);      // += B^T F  :705
      for (int idx = tid; idx < m * nn; idx += TPB)
        K[idx] = H[idx];                           // :707
      wave_sync();
      solve_lower(Gf, m, K, nn, tid);              // :708
      solve_lower_t(Gf, m, K, nn, tid);            // :710
      for (int idx = tid; idx < m * nn; idx += TPB)
        K[idx] = -K[idx];                          // :713
      wave_sync();
      gemm_tn(nn, nc, nn, A, F, S(1), V, tid);     // V += A^T F  :715
      gemm_tn(nn, m, nn, K, H, S(0), F, tid);      // F_parent = K^T H  :718
      for (int idx = tid; idx < nn * nn; idx += TPB)
        V[idx] += F[idx];                          // :719
      wave_sync();
    }
    if (result != 0)
      break;
    result = factor_F(in0 + mt.od[node], V, ws + mt.oF[node], ws + mt.osd[node],
                      ws + mt.osdi[node], nn, tid); // :722-727
  }
  if (tid == 0)
    status[b] = result;
}

// lqr.cpp:735-871, one problem per workgroup; needs a successful factor.
template <class S>
__global__ __launch_bounds__(TPB) void solve_kernel(
    const Meta mt, const S *in0_all, const S *in1_all, S *ws_all, S *gain_all,
    S *out_all, const int *status, const long batch) {
  const long b = blockIdx.x;
  if (b >= batch || status[b] != 0)
    return;
  const int tid = threadIdx.x;
  const S *in0 = in0_all + b * mt.in0_len;
  const S *in1 = in1_all + b * mt.in1_len;
  S *ws = ws_all + b * mt.ws_len;
  S *gain = gain_all + b * mt.gain_len;
  S *out = out_all + b * mt.out_len;
  S *g = ws + mt.scratch_ws + (long)mt.max_m * mt.max_n +
         (long)mt.max_n * mt.max_n;
  S *h = g + mt.max_n, *f = h + mt.max_m;

  for (int order = 0; order < mt.num_nodes; ++order) { // :738-796
    const int node = mt.postorder[order];
    const int nn = mt.state_dims[node];
    const S *q = in1 + mt.oq[node];
    S *v = ws + mt.ov[node];
    for (int i = tid; i < nn; i += TPB)
      v[i] = q[i];
    wave_sync();
    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1];
         ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      const S *A = in0 + mt.oA[e], *B = in0 + mt.oB[e], *r = in1 + mt.orr[e];
      const S *cc = in1 + mt.oc[child], *dc = in0 + mt.od[child];
      const S *W = ws + mt.oW[e], *Gf = ws + mt.oG[e], *K = gain + mt.oK[e];
      S *k = gain + mt.ok[e];
      const S *vc = ws + mt.ov[child];

      for (int i = tid; i < nc; i += TPB)
        f[i] = dc[i] * vc[i] - cc[i];              // :778-779
      wave_sync();
      for (int i = tid; i < nc; i += TPB) {        // g = v_c - W f  :780-781
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += W[i + (long)j * nc] * f[j];
        g[i] = vc[i] - s;
      }
      wave_sync();
      for (int i = tid; i < m; i += TPB) {         // h = r + B^T g  :783-784
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += B[j + (long)i * nc] * g[j];
        h[i] = r[i] + s;
        k[i] = h[i];                               // :785
      }
      wave_sync();
      solve_lower(Gf, m, k, 1, tid);
      solve_lower_t(Gf, m, k, 1, tid);
      for (int i = tid; i < m; i += TPB)
        k[i] = -k[i];                              // :791
      for (int i = tid; i < nn; i += TPB) {        // v += A^T g + K^T h  :793-794
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += A[j + (long)i * nc] * g[j];
        for (int j = 0; j < m; ++j)
          s += K[j + (long)i * m] * h[j];
        v[i] += s;
      }
      wave_sync();
    }
  }

  { // root  :798-819
    const int root = mt.preorder[0];
    const int n = mt.state_dims[root];
    const S *cr = in1 + mt.oc[root], *dr = in0 + mt.od[root];
    const S *V = ws + mt.oV[root], *Ff = ws + mt.oF[root],
            *sd = ws + mt.osd[root], *sdi = ws + mt.osdi[root],
            *v = ws + mt.ov[root];
    S *x = out + mt.ox[root], *y = out + mt.oy[root];
    for (int i = tid; i < n; i += TPB)
      f[i] = dr[i] * v[i] - cr[i];
    wave_sync();
    F_inv_mult(Ff, f, x, sd, sdi, n, tid);
    for (int i = tid; i < n; i += TPB)
      x[i] = -x[i];
    wave_sync();
    for (int i = tid; i < n; i += TPB) {
      S s = S(0);
      for (int j = 0; j < n; ++j)
        s += V[i + (long)j * n] * x[j];
      y[i] = v[i] + s;
    }
    wave_sync();
  }

  for (int order = 0; order < mt.num_nodes; ++order) { // rollout  :821-870
    const int node = mt.preorder[order];
    const int nn = mt.state_dims[node];
    const S *xn = out + mt.ox[node];
    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1];
         ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      const S *A = in0 + mt.oA[e], *B = in0 + mt.oB[e];
      const S *cc = in1 + mt.oc[child], *dc = in0 + mt.od[child];
      const S *K = gain + mt.oK[e], *k = gain + mt.ok[e];
      const S *Vc = ws + mt.oV[child], *Fc = ws + mt.oF[child],
              *sdc = ws + mt.osd[child], *sdic = ws + mt.osdi[child],
              *vc = ws + mt.ov[child];
      S *u = out + mt.ou[e];
      S *xc = out + mt.ox[child], *yc = out + mt.oy[child];

      for (int i = tid; i < m; i += TPB) {         // u = k + K x  :856-857
        S s = S(0);
        for (int j = 0; j < nn; ++j)
          s += K[i + (long)j * m] * xn[j];
        u[i] = k[i] + s;
      }
      wave_sync();
      for (int i = tid; i < nc; i += TPB) {        // :859-862
        S s = cc[i] - dc[i] * vc[i];
        for (int j = 0; j < nn; ++j)
          s += A[i + (long)j * nc] * xn[j];
        for (int j = 0; j < m; ++j)
          s += B[i + (long)j * nc] * u[j];
        f[i] = s;
      }
      wave_sync();
      F_inv_mult(Fc, f, xc, sdc, sdic, nc, tid);   // :863-865
      for (int i = tid; i < nc; i += TPB) {        // y = v + V x  :867-868
        S s = S(0);
        for (int j = 0; j < nc; ++j)
          s += Vc[i + (long)j * nc] * xc[j];
        yc[i] = vc[i] + s;
      }
      wave_sync();
    }
  }
}

} // namespace tree
} // namespace sipamd
