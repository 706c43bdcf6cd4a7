/*
 * lqr_oracle.c -- see lqr_oracle.h.  TEST INFRASTRUCTURE ONLY: a CPU
 * restatement of /root/reference/sip_optimal_control/lqr.cpp:473-871 used as
 * the parity oracle and as the "port" CPU baseline.  Never linked into the
 * product.
 *
 * Dense kernels: the reference delegates to Eigen 3.4.0 (MODULE.bazel:16):
 *   - Eigen::LLT<Ref<MatrixXd>> in place, lower (lqr.cpp:505,697): restated
 *     below as the textbook unblocked lower Cholesky (Eigen's own unblocked
 *     path, taken for sizes < 32): pivot x = a_kk - ||L_k,0:k||^2, failure iff
 *     x <= 0, only the lower triangle is read or written.
 *   - triangularView<Lower>().solveInPlace / transpose().triangularView<Upper>
 *     ().solveInPlace (lqr.cpp:517-519,542-544,708-712,786-790): forward and
 *     backward substitution.
 *   - noalias() GEMM/GEMV on dynamic Maps: plain triple loops.
 * Operation order differs from Eigen's vectorised kernels only in the order
 * of floating-point summation; nothing in the reference depends on bitwise
 * equality (its own tests are tolerance-based, tests/lqr_test.cpp:260,1000).
 */
#include "lqr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- dense helpers (column-major, ld == rows) --------------------------- */

/* In-place lower Cholesky.  Returns -1 on success, else the failing pivot. */
static int chol_lower(double *a, int n) {
  for (int k = 0; k < n; ++k) {
    double x = a[k + (long)k * n];
    for (int j = 0; j < k; ++j)
      x -= a[k + (long)j * n] * a[k + (long)j * n];
    if (x <= 0.0)
      return k; /* Eigen LLT: NumericalIssue iff a pivot is <= 0 */
    x = sqrt(x);
    a[k + (long)k * n] = x;
    for (int i = k + 1; i < n; ++i) {
      double s = a[i + (long)k * n];
      for (int j = 0; j < k; ++j)
        s -= a[i + (long)j * n] * a[k + (long)j * n];
      a[i + (long)k * n] = s / x;
    }
  }
  return -1;
}

/* X <- L^{-1} X, L n x n lower (only lower read), X n x nrhs. */
static void solve_lower(const double *L, int n, double *X, int nrhs) {
  for (int col = 0; col < nrhs; ++col) {
    double *x = X + (long)col * n;
    for (int i = 0; i < n; ++i) {
      double s = x[i];
      for (int j = 0; j < i; ++j)
        s -= L[i + (long)j * n] * x[j];
      x[i] = s / L[i + (long)i * n];
    }
  }
}

/* X <- L^{-T} X. */
static void solve_lower_transposed(const double *L, int n, double *X,
                                   int nrhs) {
  for (int col = 0; col < nrhs; ++col) {
    double *x = X + (long)col * n;
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i];
      for (int j = i + 1; j < n; ++j)
        s -= L[j + (long)i * n] * x[j];
      x[i] = s / L[i + (long)i * n];
    }
  }
}

/* C (p x r) = beta*C + A^T B with A (q x p), B (q x r). */
static void gemm_tn(int p, int q, int r, const double *A, const double *B,
                    double beta, double *C) {
  for (int j = 0; j < r; ++j)
    for (int i = 0; i < p; ++i) {
      double s = 0.0;
      for (int l = 0; l < q; ++l)
        s += A[l + (long)i * q] * B[l + (long)j * q];
      C[i + (long)j * p] = (beta == 0.0 ? 0.0 : beta * C[i + (long)j * p]) + s;
    }
}

/* C (p x r) = beta*C + A B with A (p x q), B (q x r). */
static void gemm_nn(int p, int q, int r, const double *A, const double *B,
                    double beta, double *C) {
  for (int j = 0; j < r; ++j)
    for (int i = 0; i < p; ++i) {
      double s = 0.0;
      for (int l = 0; l < q; ++l)
        s += A[i + (long)l * p] * B[l + (long)j * q];
      C[i + (long)j * p] = (beta == 0.0 ? 0.0 : beta * C[i + (long)j * p]) + s;
    }
}

/* ---- lqr.cpp:475-549: the delta-regularized building blocks ------------- */

/* lqr.cpp:475-485 */
static int compute_delta_sqrt(const double *delta, double *sd, double *sdi,
                              int n) {
  for (int i = 0; i < n; ++i) {
    if (delta[i] <= 0.0)
      return 0;
    sd[i] = sqrt(delta[i]);
    sdi[i] = 1.0 / sd[i];
  }
  return 1;
}

/* lqr.cpp:487-509: F = I + D^{1/2} V D^{1/2}, Cholesky in place. */
static int factor_F(const double *delta, const double *V, double *F, double *sd,
                    double *sdi, int n) {
  if (!compute_delta_sqrt(delta, sd, sdi, n))
    return LQR_ORACLE_INVALID_DELTA;
  for (int col = 0; col < n; ++col) {
    for (int row = 0; row < n; ++row)
      F[row + (long)col * n] = sd[row] * V[row + (long)col * n] * sd[col];
    F[col + (long)col * n] += 1.0;
  }
  return chol_lower(F, n) < 0 ? LQR_ORACLE_SUCCESS
                              : LQR_ORACLE_F_FACTORIZATION_FAILURE;
}

/* lqr.cpp:511-529: W = D^{-1/2} (I - F^{-1}) D^{-1/2}. */
static void compute_regularized_W(const double *Ffac, double *W,
                                  const double *sdi, int n) {
  memset(W, 0, sizeof(double) * (size_t)n * (size_t)n);
  for (int i = 0; i < n; ++i)
    W[i + (long)i * n] = 1.0;
  solve_lower(Ffac, n, W, n);
  solve_lower_transposed(Ffac, n, W, n);
  for (long e = 0; e < (long)n * n; ++e)
    W[e] *= -1.0;
  for (int i = 0; i < n; ++i)
    W[i + (long)i * n] += 1.0;
  for (int col = 0; col < n; ++col)
    for (int row = 0; row < n; ++row)
      W[row + (long)col * n] *= sdi[row] * sdi[col];
}

/* lqr.cpp:531-549: result = (I + Delta V)^{-1} rhs. */
static void F_inv_mult_vector(const double *Ffac, const double *rhs,
                              double *result, const double *sd,
                              const double *sdi, int n) {
  for (int i = 0; i < n; ++i)
    result[i] = sdi[i] * rhs[i];
  solve_lower(Ffac, n, result, 1);
  solve_lower_transposed(Ffac, n, result, 1);
  for (int i = 0; i < n; ++i)
    result[i] *= sd[i];
}

/* ---- workspace ---------------------------------------------------------- */

static int max_of(const int *v, int count) {
  int best = 0;
  for (int i = 0; i < count; ++i)
    if (v[i] > best)
      best = v[i];
  return best;
}

static double *alloc_d(long count) {
  return (double *)calloc((size_t)(count > 0 ? count : 1), sizeof(double));
}

static int *alloc_i(long count) {
  return (int *)calloc((size_t)(count > 0 ? count : 1), sizeof(int));
}

/* lqr.cpp:223-272.  Slot sizes follow the reference: W is max_n^2 and K is
 * m_e x max_n per edge (lqr.cpp:243-244). */
int lqr_oracle_workspace_reserve(lqr_oracle_workspace *ws,
                                 const lqr_oracle_problem *p) {
  const int E = p->num_edges, N = E + 1;
  const int max_n = max_of(p->state_dims, N);
  const int max_m = max_of(p->control_dims, E);
  memset(ws, 0, sizeof(*ws));
  ws->num_edges = E;
  ws->W = (double **)calloc((size_t)(E > 0 ? E : 1), sizeof(double *));
  ws->K = (double **)calloc((size_t)(E > 0 ? E : 1), sizeof(double *));
  ws->G_factor = (double **)calloc((size_t)(E > 0 ? E : 1), sizeof(double *));
  ws->k = (double **)calloc((size_t)(E > 0 ? E : 1), sizeof(double *));
  ws->V = (double **)calloc((size_t)N, sizeof(double *));
  ws->F_factor = (double **)calloc((size_t)N, sizeof(double *));
  ws->sqrt_delta = (double **)calloc((size_t)N, sizeof(double *));
  ws->sqrt_delta_inv = (double **)calloc((size_t)N, sizeof(double *));
  ws->v = (double **)calloc((size_t)N, sizeof(double *));
  for (int e = 0; e < E; ++e) {
    const int m = p->control_dims[e];
    ws->W[e] = alloc_d((long)max_n * max_n);
    ws->K[e] = alloc_d((long)m * max_n);
    ws->G_factor[e] = alloc_d((long)m * m);
    ws->k[e] = alloc_d(m);
  }
  for (int node = 0; node < N; ++node) {
    const int n = p->state_dims[node];
    ws->V[node] = alloc_d((long)n * n);
    ws->F_factor[node] = alloc_d((long)n * n);
    ws->sqrt_delta[node] = alloc_d(n);
    ws->sqrt_delta_inv[node] = alloc_d(n);
    ws->v[node] = alloc_d(n);
  }
  ws->G = alloc_d((long)max_m * max_m);
  ws->g = alloc_d(max_n);
  ws->H = alloc_d((long)max_m * max_n);
  ws->h = alloc_d(max_m);
  ws->F = alloc_d((long)max_n * max_n);
  ws->f = alloc_d(max_n);
  ws->child_offsets = alloc_i(N + 1);
  ws->child_edges = alloc_i(E);
  ws->edge_parents = alloc_i(E);
  ws->edge_children = alloc_i(E);
  ws->preorder_nodes = alloc_i(N);
  ws->postorder_nodes = alloc_i(N);
  ws->node_marks = alloc_i(N);
  ws->traversal_status = LQR_ORACLE_INVALID_TOPOLOGY;
  return 0;
}

void lqr_oracle_workspace_free(lqr_oracle_workspace *ws) {
  const int E = ws->num_edges, N = E + 1;
  for (int e = 0; e < E; ++e) {
    free(ws->W[e]);
    free(ws->K[e]);
    free(ws->G_factor[e]);
    free(ws->k[e]);
  }
  for (int node = 0; node < N; ++node) {
    free(ws->V[node]);
    free(ws->F_factor[node]);
    free(ws->sqrt_delta[node]);
    free(ws->sqrt_delta_inv[node]);
    free(ws->v[node]);
  }
  free(ws->W);
  free(ws->K);
  free(ws->G_factor);
  free(ws->k);
  free(ws->V);
  free(ws->F_factor);
  free(ws->sqrt_delta);
  free(ws->sqrt_delta_inv);
  free(ws->v);
  free(ws->G);
  free(ws->g);
  free(ws->H);
  free(ws->h);
  free(ws->F);
  free(ws->f);
  free(ws->child_offsets);
  free(ws->child_edges);
  free(ws->edge_parents);
  free(ws->edge_children);
  free(ws->preorder_nodes);
  free(ws->postorder_nodes);
  free(ws->node_marks);
  memset(ws, 0, sizeof(*ws));
}

/* ---- lqr.cpp:563-631: topology compilation ------------------------------ */

int lqr_oracle_compile_topology(const lqr_oracle_problem *p,
                                lqr_oracle_workspace *ws) {
  const int E = p->num_edges, N = E + 1;
  int status = LQR_ORACLE_INVALID_TOPOLOGY;
  if (p->edge_parents == NULL || p->edge_children == NULL)
    goto done;
  if (p->root < 0 || p->root >= N)
    goto done;

  /* counting sort of edges by parent -> CSR children lists (:576-598) */
  for (int i = 0; i <= N; ++i)
    ws->child_offsets[i] = 0;
  for (int e = 0; e < E; ++e) {
    const int parent = p->edge_parents[e], child = p->edge_children[e];
    if (parent < 0 || parent >= N || child < 0 || child >= N ||
        parent == child)
      goto done;
    ws->edge_parents[e] = parent;
    ws->edge_children[e] = child;
    ++ws->child_offsets[parent + 1];
  }
  for (int node = 0; node < N; ++node)
    ws->child_offsets[node + 1] += ws->child_offsets[node];
  for (int node = 0; node < N; ++node)
    ws->postorder_nodes[node] = ws->child_offsets[node]; /* cursor */
  for (int e = 0; e < E; ++e)
    ws->child_edges[ws->postorder_nodes[ws->edge_parents[e]]++] = e;

  /* iterative DFS, children pushed in reverse so the lowest edge index is
   * visited first (:600-619); postorder_nodes doubles as the stack. */
  {
    int stack = 0, visited = 0;
    ws->postorder_nodes[stack++] = p->root;
    for (int node = 0; node < N; ++node)
      ws->node_marks[node] = 0;
    while (stack > 0) {
      const int node = ws->postorder_nodes[--stack];
      if (visited >= N || ws->node_marks[node] != 0)
        goto done;
      ws->node_marks[node] = 1;
      ws->preorder_nodes[visited++] = node;
      for (int ci = ws->child_offsets[node + 1] - 1;
           ci >= ws->child_offsets[node]; --ci)
        ws->postorder_nodes[stack++] = ws->edge_children[ws->child_edges[ci]];
    }
    if (visited != N)
      goto done;
  }
  for (int order = 0; order < N; ++order)
    ws->postorder_nodes[order] = ws->preorder_nodes[N - 1 - order];
  status = LQR_ORACLE_SUCCESS;
done:
  ws->traversal_status = status;
  return status;
}

/* ---- lqr.cpp:645-731: backward matrix Riccati --------------------------- */

int lqr_oracle_factor(const lqr_oracle_problem *p, lqr_oracle_workspace *ws) {
  if (ws->traversal_status != LQR_ORACLE_SUCCESS)
    return ws->traversal_status;
  const int N = p->num_edges + 1;

  for (int order = 0; order < N; ++order) {
    const int node = ws->postorder_nodes[order];
    const int nn = p->state_dims[node];
    double *V = ws->V[node];
    memcpy(V, p->Q[node], sizeof(double) * (size_t)nn * (size_t)nn); /* :658 */

    for (int ci = ws->child_offsets[node]; ci < ws->child_offsets[node + 1];
         ++ci) {
      const int e = ws->child_edges[ci];
      const int child = ws->edge_children[e];
      const int nc = p->state_dims[child];
      const int m = p->control_dims[e];
      const double *A = p->A[e], *B = p->B[e], *M = p->M[e], *R = p->R[e];
      double *W = ws->W[e], *Gf = ws->G_factor[e], *K = ws->K[e];
      double *H = ws->H, *F = ws->F;

      compute_regularized_W(ws->F_factor[child], W, ws->sqrt_delta_inv[child],
                            nc); /* :689 */

      gemm_tn(m, nc, nc, B, W, 0.0, H); /* H_child = B^T W   :692 */
      memcpy(Gf, R, sizeof(double) * (size_t)m * (size_t)m); /* :693 */
      gemm_nn(m, nc, m, H, B, 1.0, Gf);                      /* :694 */
      if (chol_lower(Gf, m) >= 0)                            /* :696-701 */
        return LQR_ORACLE_G_FACTORIZATION_FAILURE;

      gemm_nn(nc, nc, nn, W, A, 0.0, F); /* F = W A  :703 */
      for (int col = 0; col < nn; ++col) /* H_parent = M^T  :704 */
        for (int row = 0; row < m; ++row)
          H[row + (long)col * m] = M[col + (long)row * nn];
      gemm_tn(m, nc, nn, B, F, 1.0, H); /* += B^T F  :705 */

      memcpy(K, H, sizeof(double) * (size_t)m * (size_t)nn); /* :707 */
      solve_lower(Gf, m, K, nn);                             /* :708 */
      solve_lower_transposed(Gf, m, K, nn);                  /* :710 */
      for (long i = 0; i < (long)m * nn; ++i)
        K[i] *= -1.0; /* :713 */

      gemm_tn(nn, nc, nn, A, F, 1.0, V); /* V += A^T F  :715 */
      gemm_tn(nn, m, nn, K, H, 0.0, F);  /* F_parent = K^T H  :718 */
      for (long i = 0; i < (long)nn * nn; ++i)
        V[i] += F[i]; /* :719 */
    }

    const int st = factor_F(p->delta[node], V, ws->F_factor[node],
                            ws->sqrt_delta[node], ws->sqrt_delta_inv[node],
                            nn); /* :722-727 */
    if (st != LQR_ORACLE_SUCCESS)
      return st;
  }
  return LQR_ORACLE_SUCCESS;
}

/* ---- lqr.cpp:735-871: affine backward sweep, root solve, rollout -------- */

void lqr_oracle_solve(const lqr_oracle_problem *p, lqr_oracle_workspace *ws,
                      double **x, double **u, double **y) {
  const int N = p->num_edges + 1;

  for (int order = 0; order < N; ++order) { /* :738-796 */
    const int node = ws->postorder_nodes[order];
    const int nn = p->state_dims[node];
    double *v = ws->v[node];
    memcpy(v, p->q[node], sizeof(double) * (size_t)nn);

    for (int ci = ws->child_offsets[node]; ci < ws->child_offsets[node + 1];
         ++ci) {
      const int e = ws->child_edges[ci];
      const int child = ws->edge_children[e];
      const int nc = p->state_dims[child];
      const int m = p->control_dims[e];
      const double *A = p->A[e], *B = p->B[e];
      const double *vc = ws->v[child], *W = ws->W[e];
      double *g = ws->g, *h = ws->h, *f = ws->f, *k = ws->k[e];

      for (int i = 0; i < nc; ++i)
        f[i] = p->delta[child][i] * vc[i] - p->c[child][i]; /* :778-779 */
      for (int i = 0; i < nc; ++i) {                        /* :780-781 */
        double s = 0.0;
        for (int j = 0; j < nc; ++j)
          s += W[i + (long)j * nc] * f[j];
        g[i] = vc[i] - s;
      }
      for (int i = 0; i < m; ++i) { /* h = r + B^T g  :783-784 */
        double s = 0.0;
        for (int j = 0; j < nc; ++j)
          s += B[j + (long)i * nc] * g[j];
        h[i] = p->r[e][i] + s;
      }
      memcpy(k, h, sizeof(double) * (size_t)m); /* :785 */
      solve_lower(ws->G_factor[e], m, k, 1);
      solve_lower_transposed(ws->G_factor[e], m, k, 1);
      for (int i = 0; i < m; ++i)
        k[i] *= -1.0; /* :791 */

      for (int i = 0; i < nn; ++i) { /* v += A^T g + K^T h  :793-794 */
        double s = 0.0;
        for (int j = 0; j < nc; ++j)
          s += A[j + (long)i * nc] * g[j];
        for (int j = 0; j < m; ++j)
          s += ws->K[e][j + (long)i * m] * h[j];
        v[i] += s;
      }
    }
  }

  { /* root: :798-819 */
    const int root = ws->preorder_nodes[0];
    const int n = p->state_dims[root];
    double *f = ws->f;
    for (int i = 0; i < n; ++i)
      f[i] = p->delta[root][i] * ws->v[root][i] - p->c[root][i];
    F_inv_mult_vector(ws->F_factor[root], f, x[root], ws->sqrt_delta[root],
                      ws->sqrt_delta_inv[root], n);
    for (int i = 0; i < n; ++i)
      x[root][i] *= -1.0;
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < n; ++j)
        s += ws->V[root][i + (long)j * n] * x[root][j];
      y[root][i] = ws->v[root][i] + s;
    }
  }

  for (int order = 0; order < N; ++order) { /* rollout: :821-870 */
    const int node = ws->preorder_nodes[order];
    const int nn = p->state_dims[node];
    const double *xn = x[node];
    for (int ci = ws->child_offsets[node]; ci < ws->child_offsets[node + 1];
         ++ci) {
      const int e = ws->child_edges[ci];
      const int child = ws->edge_children[e];
      const int nc = p->state_dims[child];
      const int m = p->control_dims[e];
      const double *A = p->A[e], *B = p->B[e], *K = ws->K[e];
      double *f = ws->f;

      for (int i = 0; i < m; ++i) { /* u = k + K x  :856-857 */
        double s = 0.0;
        for (int j = 0; j < nn; ++j)
          s += K[i + (long)j * m] * xn[j];
        u[e][i] = ws->k[e][i] + s;
      }
      for (int i = 0; i < nc; ++i) { /* :859-862 */
        double s = p->c[child][i] - p->delta[child][i] * ws->v[child][i];
        for (int j = 0; j < nn; ++j)
          s += A[i + (long)j * nc] * xn[j];
        for (int j = 0; j < m; ++j)
          s += B[i + (long)j * nc] * u[e][j];
        f[i] = s;
      }
      F_inv_mult_vector(ws->F_factor[child], f, x[child],
                        ws->sqrt_delta[child], ws->sqrt_delta_inv[child],
                        nc); /* :863-865 */
      for (int i = 0; i < nc; ++i) { /* y = v + V x  :867-868 */
        double s = 0.0;
        for (int j = 0; j < nc; ++j)
          s += ws->V[child][i + (long)j * nc] * x[child][j];
        y[child][i] = ws->v[child][i] + s;
      }
    }
  }
}

/* ---- packed uniform-chain batch (test / baseline convenience) ----------- */

long lqr_oracle_chain_mats_len(int n, int m, int T) {
  return (long)(T + 1) * ((long)n * n + n) +
         (long)T * ((long)n * n + 2L * n * m + (long)m * m);
}
long lqr_oracle_chain_vecs_len(int n, int m, int T) {
  return (long)(T + 1) * 2 * n + (long)T * m;
}
long lqr_oracle_chain_gains_len(int n, int m, int T) {
  return (long)T * ((long)m * n + m);
}

typedef struct {
  lqr_oracle_problem prob;
  lqr_oracle_workspace ws;
  int *ints; /* parents | children | state_dims | control_dims */
  double **ptrs;
  double **x, **u, **y;
} chain_ctx;

static void chain_ctx_init(chain_ctx *c, int n, int m, int T) {
  const int N = T + 1;
  c->ints = (int *)malloc(sizeof(int) * (size_t)(2 * T + N + T + 4));
  int *parents = c->ints, *children = parents + T, *sdims = children + T,
      *cdims = sdims + N;
  for (int e = 0; e < T; ++e) { /* Topology::set_chain, lqr.cpp:32-40 */
    parents[e] = e;
    children[e] = e + 1;
    cdims[e] = m;
  }
  for (int i = 0; i < N; ++i)
    sdims[i] = n;
  c->ptrs = (double **)calloc((size_t)(4 * N + 5 * T + 2 * N + T + 8),
                              sizeof(double *));
  double **cur = c->ptrs;
  c->prob.Q = cur, cur += N;
  c->prob.q = cur, cur += N;
  c->prob.c = cur, cur += N;
  c->prob.delta = cur, cur += N;
  c->prob.M = cur, cur += T;
  c->prob.R = cur, cur += T;
  c->prob.r = cur, cur += T;
  c->prob.A = cur, cur += T;
  c->prob.B = cur, cur += T;
  c->x = cur, cur += N;
  c->y = cur, cur += N;
  c->u = cur, cur += T;
  c->prob.num_edges = T;
  c->prob.root = 0;
  c->prob.edge_parents = parents;
  c->prob.edge_children = children;
  c->prob.state_dims = sdims;
  c->prob.control_dims = cdims;
  lqr_oracle_workspace_reserve(&c->ws, &c->prob);
  lqr_oracle_compile_topology(&c->prob, &c->ws);
}

static void chain_ctx_free(chain_ctx *c) {
  lqr_oracle_workspace_free(&c->ws);
  free(c->ints);
  free(c->ptrs);
}

static void chain_solve_one(chain_ctx *c, int n, int m, int T,
                            const double *mats, const double *vecs, double *sol,
                            double *gains, int *status) {
  /* The reference takes non-const double** (lqr.hpp:76-85); it only reads. */
  double *mp = (double *)mats, *vp = (double *)vecs, *sp = sol;
  for (int i = 0; i <= T; ++i) {
    c->prob.Q[i] = mp, mp += (long)n * n;
    c->prob.delta[i] = mp, mp += n;
    c->prob.q[i] = vp, vp += n;
    c->prob.c[i] = vp, vp += n;
    c->x[i] = sp, sp += n;
    c->y[i] = sp, sp += n;
    if (i < T) {
      c->prob.A[i] = mp, mp += (long)n * n;
      c->prob.B[i] = mp, mp += (long)n * m;
      c->prob.M[i] = mp, mp += (long)n * m;
      c->prob.R[i] = mp, mp += (long)m * m;
      c->prob.r[i] = vp, vp += m;
      c->u[i] = sp, sp += m;
    }
  }
  *status = lqr_oracle_factor(&c->prob, &c->ws);
  if (*status != LQR_ORACLE_SUCCESS)
    return;
  lqr_oracle_solve(&c->prob, &c->ws, c->x, c->u, c->y);
  if (gains != NULL) {
    for (int e = 0; e < T; ++e) {
      memcpy(gains, c->ws.K[e], sizeof(double) * (size_t)m * (size_t)n);
      gains += (long)m * n;
      memcpy(gains, c->ws.k[e], sizeof(double) * (size_t)m);
      gains += m;
    }
  }
}

int lqr_oracle_chain_batch(int n, int m, int T, long batch, const double *mats,
                           const double *vecs, double *sol, double *gains,
                           int *status, int threads) {
  const long ml = lqr_oracle_chain_mats_len(n, m, T);
  const long vl = lqr_oracle_chain_vecs_len(n, m, T);
  const long gl = lqr_oracle_chain_gains_len(n, m, T);
  if (threads < 1)
    threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
  {
    chain_ctx ctx;
    memset(&ctx, 0, sizeof(ctx));
    chain_ctx_init(&ctx, n, m, T);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long p = 0; p < batch; ++p)
      chain_solve_one(&ctx, n, m, T, mats + p * ml, vecs + p * vl,
                      sol + p * vl, gains ? gains + p * gl : NULL, status + p);
    chain_ctx_free(&ctx);
  }
  return 0;
}
