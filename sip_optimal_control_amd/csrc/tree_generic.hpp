// tree_generic.hpp -- reference-shaped regularized tree-LQR on the GPU: any tree
// topology, per-node state / per-edge control dimensions (0 allowed), fp64.
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
// problems of the batch).  Offsets are in scalars from the start of a
// problem's input / workspace / output arena.
struct Meta {
  int num_edges, num_nodes, root, max_n, max_m;
  const int *state_dims, *control_dims;
  const int *edge_parents, *edge_children;
  const int *child_offsets, *child_edges, *preorder, *postorder;
  const long *node_in, *edge_in;   // input arena
  const long *node_ws, *edge_ws;   // workspace arena
  const long *node_out, *edge_out; // output arena
  long scratch_ws;                 // H | F | g | h | f
  long in_len, ws_len, out_len;
};

// input arena, node block:  Q (n*n) | q (n) | c (n) | delta (n)
// input arena, edge block:  A (nc*np) | B (nc*m) | M (np*m) | R (m*m) | r (m)
// workspace, edge block:    W (max_n^2) | K (m*np) | G_factor (m*m) | k (m)
// workspace, node block:    V (n*n) | F_factor (n*n) | sqrt_delta (n) |
//                           sqrt_delta_inv (n) | v (n)
// output arena: node block x (n) | y (n); edge block u (m)

constexpr int TPB = 64;

__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// In-place lower Cholesky (Eigen LLT unblocked).  Uniform return: failing
// pivot index or -1.
__device__ inline int chol_lower(double *a, const int n, const int tid) {
  for (int k = 0; k < n; ++k) {
    double x = a[k + (long)k * n];
    for (int j = 0; j < k; ++j)
      x -= a[k + (long)j * n] * a[k + (long)j * n];
    if (x <= 0.0)
      return k;
    const double d = sqrt(x);
    for (int i = k + 1 + tid; i < n; i += TPB) {
      double s = a[i + (long)k * n];
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
__device__ inline void solve_lower(const double *L, const int n, double *X,
                                   const int nrhs, const int tid) {
  for (int col = tid; col < nrhs; col += TPB) {
    double *x = X + (long)col * n;
    for (int i = 0; i < n; ++i) {
      double s = x[i];
      for (int j = 0; j < i; ++j)
        s -= L[i + (long)j * n] * x[j];
      x[i] = s / L[i + (long)i * n];
    }
  }
  wave_sync();
}

// X <- L^{-T} X.
__device__ inline void solve_lower_t(const double *L, const int n, double *X,
                                     const int nrhs, const int tid) {
  for (int col = tid; col < nrhs; col += TPB) {
    double *x = X + (long)col * n;
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i];
      for (int j = i + 1; j < n; ++j)
        s -= L[j + (long)i * n] * x[j];
      x[i] = s / L[i + (long)i * n];
    }
  }
  wave_sync();
}

// C (p x r) = beta C + A^T B, A (q x p), B (q x r).
__device__ inline void gemm_tn(const int p, const int q, const int r,
                               const double *A, const double *B,
                               const double beta, double *C, const int tid) {
  for (int idx = tid; idx < p * r; idx += TPB) {
    const int i = idx % p, j = idx / p;
    double s = 0.0;
    for (int l = 0; l < q; ++l)
      s += A[l + (long)i * q] * B[l + (long)j * q];
    C[idx] = (beta == 0.0 ? 0.0 : beta * C[idx]) + s;
  }
  wave_sync();
}

// C (p x r) = beta C + A B, A (p x q), B (q x r).
__device__ inline void gemm_nn(const int p, const int q, const int r,
                               const double *A, const double *B,
                               const double beta, double *C, const int tid) {
  for (int idx = tid; idx < p * r; idx += TPB) {
    const int i = idx % p, j = idx / p;
    double s = 0.0;
    for (int l = 0; l < q; ++l)
      s += A[i + (long)l * p] * B[l + (long)j * q];
    C[idx] = (beta == 0.0 ? 0.0 : beta * C[idx]) + s;
  }
  wave_sync();
}

// lqr.cpp:487-509.  Returns 0 / INVALID_DELTA(1) / F_FACTORIZATION_FAILURE(2).
__device__ inline int factor_F(const double *delta, const double *V, double *F,
                               double *sd, double *sdi, const int n,
                               const int tid) {
  int bad = 0;
  for (int i = 0; i < n; ++i) // uniform scan, as the reference's early return
    if (delta[i] <= 0.0) {
      bad = 1;
      break;
    }
  if (bad)
    return 1;
  for (int i = tid; i < n; i += TPB) {
    sd[i] = sqrt(delta[i]);
    sdi[i] = 1.0 / sd[i];
  }
  wave_sync();
  for (int idx = tid; idx < n * n; idx += TPB) {
    const int row = idx % n, col = idx / n;
    F[idx] = sd[row] * V[idx] * sd[col] + (row == col ? 1.0 : 0.0);
  }
  wave_sync();
  return chol_lower(F, n, tid) < 0 ? 0 : 2;
}

// lqr.cpp:511-529.
__device__ inline void regularized_W(const double *Ffac, double *W,
                                     const double *sdi, const int n,
                                     const int tid) {
  for (int idx = tid; idx < n * n; idx += TPB)
    W[idx] = (idx % n == idx / n) ? 1.0 : 0.0;
  wave_sync();
  solve_lower(Ffac, n, W, n, tid);
  solve_lower_t(Ffac, n, W, n, tid);
  for (int idx = tid; idx < n * n; idx += TPB) {
    const int row = idx % n, col = idx / n;
    double w = -W[idx];
    if (row == col)
      w += 1.0;
    W[idx] = w * (sdi[row] * sdi[col]);
  }
  wave_sync();
}

// lqr.cpp:531-549.
__device__ inline void F_inv_mult(const double *Ffac, const double *rhs,
                                  double *res, const double *sd,
                                  const double *sdi, const int n,
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
__global__ __launch_bounds__(TPB) void factor_kernel(const Meta mt,
                                                     const double *in_all,
                                                     double *ws_all,
                                                     int *status,
                                                     const long batch) {
  const long b = blockIdx.x;
  if (b >= batch)
    return;
  const int tid = threadIdx.x;
  const double *in = in_all + b * mt.in_len;
  double *ws = ws_all + b * mt.ws_len;
  double *H = ws + mt.scratch_ws;
  double *F = H + (long)mt.max_m * mt.max_n;
  int result = 0;

  for (int order = 0; order < mt.num_nodes && result == 0; ++order) {
    const int node = mt.postorder[order];
    const int nn = mt.state_dims[node];
    const double *nin = in + mt.node_in[node];
    double *nws = ws + mt.node_ws[node];
    double *V = nws;
    for (int idx = tid; idx < nn * nn; idx += TPB)
      V[idx] = nin[idx]; // V = Q  (lqr.cpp:658)
    wave_sync();

    for (int ci = mt.child_offsets[node];
         ci < mt.child_offsets[node + 1] && result == 0; ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      const double *ein = in + mt.edge_in[e];
      const double *A = ein, *B = A + (long)nc * nn, *M = B + (long)nc * m,
                   *R = M + (long)nn * m;
      double *ews = ws + mt.edge_ws[e];
      double *W = ews, *K = W + (long)mt.max_n * mt.max_n,
             *Gf = K + (long)m * nn;
      const double *cws = ws + mt.node_ws[child];
      const double *Fc = cws + (long)nc * nc;
      const double *sdic = Fc + (long)nc * nc + nc;

      regularized_W(Fc, W, sdic, nc, tid);      // :689
      gemm_tn(m, nc, nc, B, W, 0.0, H, tid);    // H_child = B^T W  :692
      for (int idx = tid; idx < m * m; idx += TPB)
        Gf[idx] = R[idx];                       // :693
      wave_sync();
      gemm_nn(m, nc, m, H, B, 1.0, Gf, tid);    // :694
      if (chol_lower(Gf, m, tid) >= 0) {        // :696-701
        result = 3;
        break;
      }
      gemm_nn(nc, nc, nn, W, A, 0.0, F, tid);   // F = W A  :703
      for (int idx = tid; idx < m * nn; idx += TPB) {
        const int row = idx % m, col = idx / m;
        H[idx] = M[col + (long)row * nn];       // H_parent = M^T  :704
      }
      wave_sync();
      gemm_tn(m, nc, nn, B, F, 1.0, H, tid);    // += B^T F  :705
      for (int idx = tid; idx < m * nn; idx += TPB)
        K[idx] = H[idx];                        // :707
      wave_sync();
      solve_lower(Gf, m, K, nn, tid);           // :708
      solve_lower_t(Gf, m, K, nn, tid);         // :710
      for (int idx = tid; idx < m * nn; idx += TPB)
        K[idx] = -K[idx];                       // :713
      wave_sync();
      gemm_tn(nn, nc, nn, A, F, 1.0, V, tid);   // V += A^T F  :715
      gemm_tn(nn, m, nn, K, H, 0.0, F, tid);    // F_parent = K^T H  :718
      for (int idx = tid; idx < nn * nn; idx += TPB)
        V[idx] += F[idx];                       // :719
      wave_sync();
    }
    if (result != 0)
      break;
    double *Ff = V + (long)nn * nn, *sd = Ff + (long)nn * nn, *sdi = sd + nn;
    result = factor_F(nin + (long)nn * nn + 2 * nn, V, Ff, sd, sdi, nn,
                      tid);                     // :722-727
  }
  if (tid == 0)
    status[b] = result;
}

// lqr.cpp:735-871, one problem per workgroup; needs a successful factor.
__global__ __launch_bounds__(TPB) void solve_kernel(const Meta mt,
                                                    const double *in_all,
                                                    double *ws_all,
                                                    double *out_all,
                                                    const int *status,
                                                    const long batch) {
  const long b = blockIdx.x;
  if (b >= batch || status[b] != 0)
    return;
  const int tid = threadIdx.x;
  const double *in = in_all + b * mt.in_len;
  double *ws = ws_all + b * mt.ws_len;
  double *out = out_all + b * mt.out_len;
  double *g = ws + mt.scratch_ws + (long)mt.max_m * mt.max_n +
              (long)mt.max_n * mt.max_n;
  double *h = g + mt.max_n, *f = h + mt.max_m;

  for (int order = 0; order < mt.num_nodes; ++order) { // :738-796
    const int node = mt.postorder[order];
    const int nn = mt.state_dims[node];
    const double *nin = in + mt.node_in[node];
    double *nws = ws + mt.node_ws[node];
    double *v = nws + 2L * nn * nn + 2 * nn;
    for (int i = tid; i < nn; i += TPB)
      v[i] = nin[(long)nn * nn + i]; // v = q
    wave_sync();
    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1];
         ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      const double *ein = in + mt.edge_in[e];
      const double *A = ein, *B = A + (long)nc * nn,
                   *r = B + (long)nc * m + (long)nn * m + (long)m * m;
      const double *cin = in + mt.node_in[child];
      const double *cc = cin + (long)nc * nc + nc, *dc = cc + nc;
      double *ews = ws + mt.edge_ws[e];
      const double *W = ews, *K = W + (long)mt.max_n * mt.max_n,
                   *Gf = K + (long)m * nn;
      double *k = ews + (long)mt.max_n * mt.max_n + (long)m * nn + (long)m * m;
      const double *vc = ws + mt.node_ws[child] + 2L * nc * nc + 2 * nc;

      for (int i = tid; i < nc; i += TPB)
        f[i] = dc[i] * vc[i] - cc[i];           // :778-779
      wave_sync();
      for (int i = tid; i < nc; i += TPB) {     // g = v_c - W f  :780-781
        double s = 0.0;
        for (int j = 0; j < nc; ++j)
          s += W[i + (long)j * nc] * f[j];
        g[i] = vc[i] - s;
      }
      wave_sync();
      for (int i = tid; i < m; i += TPB) {      // h = r + B^T g  :783-784
        double s = 0.0;
        for (int j = 0; j < nc; ++j)
          s += B[j + (long)i * nc] * g[j];
        h[i] = r[i] + s;
        k[i] = h[i];                            // :785
      }
      wave_sync();
      solve_lower(Gf, m, k, 1, tid);
      solve_lower_t(Gf, m, k, 1, tid);
      for (int i = tid; i < m; i += TPB)
        k[i] = -k[i];                           // :791
      for (int i = tid; i < nn; i += TPB) {     // v += A^T g + K^T h  :793-794
        double s = 0.0;
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
    const double *nin = in + mt.node_in[root];
    const double *cr = nin + (long)n * n + n, *dr = cr + n;
    const double *nws = ws + mt.node_ws[root];
    const double *V = nws, *Ff = V + (long)n * n, *sd = Ff + (long)n * n,
                 *sdi = sd + n, *v = sdi + n;
    double *x = out + mt.node_out[root], *y = x + n;
    for (int i = tid; i < n; i += TPB)
      f[i] = dr[i] * v[i] - cr[i];
    wave_sync();
    F_inv_mult(Ff, f, x, sd, sdi, n, tid);
    for (int i = tid; i < n; i += TPB)
      x[i] = -x[i];
    wave_sync();
    for (int i = tid; i < n; i += TPB) {
      double s = 0.0;
      for (int j = 0; j < n; ++j)
        s += V[i + (long)j * n] * x[j];
      y[i] = v[i] + s;
    }
    wave_sync();
  }

  for (int order = 0; order < mt.num_nodes; ++order) { // rollout  :821-870
    const int node = mt.preorder[order];
    const int nn = mt.state_dims[node];
    const double *xn = out + mt.node_out[node];
    for (int ci = mt.child_offsets[node]; ci < mt.child_offsets[node + 1];
         ++ci) {
      const int e = mt.child_edges[ci];
      const int child = mt.edge_children[e];
      const int nc = mt.state_dims[child];
      const int m = mt.control_dims[e];
      const double *ein = in + mt.edge_in[e];
      const double *A = ein, *B = A + (long)nc * nn;
      const double *cin = in + mt.node_in[child];
      const double *cc = cin + (long)nc * nc + nc, *dc = cc + nc;
      const double *ews = ws + mt.edge_ws[e];
      const double *K = ews + (long)mt.max_n * mt.max_n;
      const double *k = K + (long)m * nn + (long)m * m;
      const double *cws = ws + mt.node_ws[child];
      const double *Vc = cws, *Fc = Vc + (long)nc * nc,
                   *sdc = Fc + (long)nc * nc, *sdic = sdc + nc,
                   *vc = sdic + nc;
      double *u = out + mt.edge_out[e];
      double *xc = out + mt.node_out[child], *yc = xc + nc;

      for (int i = tid; i < m; i += TPB) {      // u = k + K x  :856-857
        double s = 0.0;
        for (int j = 0; j < nn; ++j)
          s += K[i + (long)j * m] * xn[j];
        u[i] = k[i] + s;
      }
      wave_sync();
      for (int i = tid; i < nc; i += TPB) {     // :859-862
        double s = cc[i] - dc[i] * vc[i];
        for (int j = 0; j < nn; ++j)
          s += A[i + (long)j * nc] * xn[j];
        for (int j = 0; j < m; ++j)
          s += B[i + (long)j * nc] * u[j];
        f[i] = s;
      }
      wave_sync();
      F_inv_mult(Fc, f, xc, sdc, sdic, nc, tid); // :863-865
      for (int i = tid; i < nc; i += TPB) {     // y = v + V x  :867-868
        double s = 0.0;
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
