/*
 * kkt_oracle.c -- see kkt_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Every function cites the lines of /root/reference/sip_optimal_control it
 * follows.  Loop nests keep the reference's order of floating-point
 * accumulation (constraint-major rank-1 updates, edges in index order).
 */
#include "kkt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lqr_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

struct kkt_oracle {
  int E, N, root;
  int *parents, *children, *sd, *cd, *ncd, *ngd, *ecd, *egd;
  long *off[KKT_NUM_BLOCKS];
  long model_len;
  int *voff[7];
  int x_dim, y_dim, z_dim;
  int input_valid;
  /* Workspace::RegularizedLQRData, types.hpp:163-176 */
  double **Q_mod, **M_mod, **R_mod, **q_mod, **r_mod, **c_mod, **dyn_r2;
  double **node_c_r2_inv, **edge_c_r2_inv, **node_mod_w_inv, **edge_mod_w_inv;
  double **A, **B; /* Workspace::ddyn_dx / ddyn_du, types.cpp:624-628 */
  double **x, **u, **y;
  lqr_oracle_problem prob;
  lqr_oracle_workspace ws;
  /* theta (helpers.cpp:190-240, 372-407, 896-951) */
  int p;
  long *toff[KKT_TH_NUM_BLOCKS];
  long theta_len;
  double *J_theta, *K_inv_J_theta, *S_factor, *theta_rhs, *stagewise_rhs, *stagewise_sol;
};

static int *dup_ints(const int *src, int count) {
  int *dst = (int *)calloc((size_t)(count > 0 ? count : 1), sizeof(int));
  if (src != NULL && count > 0)
    memcpy(dst, src, (size_t)count * sizeof(int));
  return dst;
}

static double **alloc_blocks(int count, const int *sizes) {
  double **tab = (double **)calloc((size_t)(count > 0 ? count : 1), sizeof(double *));
  for (int i = 0; i < count; ++i)
    tab[i] = (double *)calloc((size_t)(sizes[i] > 0 ? sizes[i] : 1), sizeof(double));
  return tab;
}

static void free_blocks(double **tab, int count) {
  if (tab == NULL)
    return;
  for (int i = 0; i < count; ++i)
    free(tab[i]);
  free(tab);
}

/* validate_input, types.cpp:68-127 (theta_dim == 0) */
static int validate(const kkt_oracle *o, int have_topology) {
  const int E = o->E, N = o->N;
  if (E < 0)
    return 0;
  for (int i = 0; i < N; ++i)
    if (o->sd[i] < 0 || o->ncd[i] < 0 || o->ngd[i] < 0)
      return 0;
  for (int e = 0; e < E; ++e)
    if (o->cd[e] < 0 || o->ecd[e] < 0 || o->egd[e] < 0)
      return 0;
  if (!have_topology || o->root < 0 || o->root >= N)
    return 0;
  for (int e = 0; e < E; ++e) {
    const int p = o->parents[e], c = o->children[e];
    if (p < 0 || p >= N || c < 0 || c >= N || p == c)
      return 0;
  }
  return 1; /* in-degree / connectivity: latched by the LQR traversal below */
}

kkt_oracle *kkt_oracle_create(int num_edges, int root, const int *edge_parents,
                              const int *edge_children, const int *state_dims,
                              const int *control_dims, const int *node_c_dims,
                              const int *node_g_dims, const int *edge_c_dims,
                              const int *edge_g_dims) {
  kkt_oracle *o = (kkt_oracle *)calloc(1, sizeof(kkt_oracle));
  const int E = num_edges < 0 ? 0 : num_edges, N = E + 1;
  o->E = num_edges, o->N = N, o->root = root;
  o->parents = dup_ints(edge_parents, E), o->children = dup_ints(edge_children, E);
  o->sd = dup_ints(state_dims, N), o->cd = dup_ints(control_dims, E);
  o->ncd = dup_ints(node_c_dims, N), o->ngd = dup_ints(node_g_dims, N);
  o->ecd = dup_ints(edge_c_dims, E), o->egd = dup_ints(edge_g_dims, E);
  o->input_valid = validate(o, edge_parents != NULL && edge_children != NULL);
  for (int b = 0; b < KKT_NUM_BLOCKS; ++b)
    o->off[b] = (long *)calloc((size_t)N, sizeof(long));
  for (int t = 0; t < 7; ++t)
    o->voff[t] = (int *)calloc((size_t)N, sizeof(int));
  if (!o->input_valid)
    return o;

  /* model arena: node i, then edge i */
  long at = 0;
  for (int i = 0; i < N; ++i) {
    const long n = o->sd[i];
    o->off[KKT_NODE_D2L_DX2][i] = at, at += n * n;
    o->off[KKT_NODE_DC_DX][i] = at, at += o->ncd[i] * n;
    o->off[KKT_NODE_DG_DX][i] = at, at += o->ngd[i] * n;
    if (i < E) {
      const int e = i;
      const long np = o->sd[o->parents[e]], nc = o->sd[o->children[e]], m = o->cd[e];
      o->off[KKT_EDGE_D2L_DX2][e] = at, at += np * np;
      o->off[KKT_EDGE_D2L_DXDU][e] = at, at += np * m;
      o->off[KKT_EDGE_D2L_DU2][e] = at, at += m * m;
      o->off[KKT_EDGE_DDYN_DX][e] = at, at += nc * np;
      o->off[KKT_EDGE_DDYN_DU][e] = at, at += nc * m;
      o->off[KKT_EDGE_DC_DX][e] = at, at += o->ecd[e] * np;
      o->off[KKT_EDGE_DC_DU][e] = at, at += o->ecd[e] * m;
      o->off[KKT_EDGE_DG_DX][e] = at, at += o->egd[e] * np;
      o->off[KKT_EDGE_DG_DU][e] = at, at += o->egd[e] * m;
    }
  }
  o->model_len = at;

  /* populate_workspace_metadata, types.cpp:24-64 */
  int xo = 0, yo = 0, zo = 0;
  for (int i = 0; i < N; ++i) {
    o->voff[KKT_X_STATE][i] = xo;
    if (i < E) {
      xo += o->sd[i];
      o->voff[KKT_X_CONTROL][i] = xo;
      xo += o->cd[i];
    }
  }
  xo = o->sd[E];
  for (int e = 0; e < E; ++e)
    xo += o->sd[e] + o->cd[e]; /* get_stagewise_x_dim, lqr.cpp:146-151 */
  for (int i = 0; i < N; ++i) {
    o->voff[KKT_Y_DYN][i] = yo, yo += o->sd[i];
    o->voff[KKT_Y_NODE_C][i] = yo, yo += o->ncd[i];
  }
  for (int e = 0; e < E; ++e)
    o->voff[KKT_Y_EDGE_C][e] = yo, yo += o->ecd[e];
  for (int i = 0; i < N; ++i)
    o->voff[KKT_Z_NODE][i] = zo, zo += o->ngd[i];
  for (int e = 0; e < E; ++e)
    o->voff[KKT_Z_EDGE][e] = zo, zo += o->egd[e];
  o->x_dim = xo, o->y_dim = yo, o->z_dim = zo;

  /* RegularizedLQRData::reserve, types.cpp (per node n*n + 3n + c + g, per
   * edge max_n*m + m*m + m + c + g) */
  int *sz = (int *)calloc((size_t)N, sizeof(int));
  int max_n = 0;
  for (int i = 0; i < N; ++i)
    max_n = o->sd[i] > max_n ? o->sd[i] : max_n;
#define NODE_BLOCKS(field, expr)   \
  for (int i = 0; i < N; ++i)      \
    sz[i] = (expr);                \
  o->field = alloc_blocks(N, sz)
#define EDGE_BLOCKS(field, expr)   \
  for (int e = 0; e < E; ++e)      \
    sz[e] = (expr);                \
  o->field = alloc_blocks(E, sz)
  NODE_BLOCKS(Q_mod, o->sd[i] * o->sd[i]);
  NODE_BLOCKS(q_mod, o->sd[i]);
  NODE_BLOCKS(c_mod, o->sd[i]);
  NODE_BLOCKS(dyn_r2, o->sd[i]);
  NODE_BLOCKS(node_c_r2_inv, o->ncd[i]);
  NODE_BLOCKS(node_mod_w_inv, o->ngd[i]);
  EDGE_BLOCKS(M_mod, max_n * o->cd[e]);
  EDGE_BLOCKS(R_mod, o->cd[e] * o->cd[e]);
  EDGE_BLOCKS(r_mod, o->cd[e]);
  EDGE_BLOCKS(edge_c_r2_inv, o->ecd[e]);
  EDGE_BLOCKS(edge_mod_w_inv, o->egd[e]);
#undef NODE_BLOCKS
#undef EDGE_BLOCKS
  free(sz);
  o->A = (double **)calloc((size_t)N, sizeof(double *));
  o->B = (double **)calloc((size_t)N, sizeof(double *));
  o->x = (double **)calloc((size_t)N, sizeof(double *));
  o->u = (double **)calloc((size_t)N, sizeof(double *));
  o->y = (double **)calloc((size_t)N, sizeof(double *));

  /* CallbackProvider ctor, helpers.cpp:11-26: the LQR solver is built once
   * with null data pointers that factor()/solve() patch before each call. */
  lqr_oracle_problem *p = &o->prob;
  p->num_edges = E, p->root = root;
  p->edge_parents = o->parents, p->edge_children = o->children;
  p->state_dims = o->sd, p->control_dims = o->cd;
  lqr_oracle_workspace_reserve(&o->ws, p);
  lqr_oracle_compile_topology(p, &o->ws);
  return o;
}

void kkt_oracle_destroy(kkt_oracle *o) {
  if (o == NULL)
    return;
  for (int b = 0; b < KKT_TH_NUM_BLOCKS; ++b)
    free(o->toff[b]);
  free(o->J_theta), free(o->K_inv_J_theta), free(o->S_factor), free(o->theta_rhs);
  free(o->stagewise_rhs), free(o->stagewise_sol);
  const int E = o->E < 0 ? 0 : o->E, N = o->N;
  if (o->input_valid) {
    lqr_oracle_workspace_free(&o->ws);
    free_blocks(o->Q_mod, N), free_blocks(o->q_mod, N), free_blocks(o->c_mod, N);
    free_blocks(o->dyn_r2, N), free_blocks(o->node_c_r2_inv, N), free_blocks(o->node_mod_w_inv, N);
    free_blocks(o->M_mod, E), free_blocks(o->R_mod, E), free_blocks(o->r_mod, E);
    free_blocks(o->edge_c_r2_inv, E), free_blocks(o->edge_mod_w_inv, E);
    free(o->A), free(o->B), free(o->x), free(o->u), free(o->y);
  }
  for (int b = 0; b < KKT_NUM_BLOCKS; ++b)
    free(o->off[b]);
  for (int t = 0; t < 7; ++t)
    free(o->voff[t]);
  free(o->parents), free(o->children), free(o->sd), free(o->cd);
  free(o->ncd), free(o->ngd), free(o->ecd), free(o->egd);
  free(o);
}

long kkt_oracle_dim(const kkt_oracle *o, int which) {
  switch (which) {
  case 0: return o->x_dim;
  case 1: return o->y_dim;
  case 2: return o->z_dim;
  case 3: return o->model_len;
  default: return -1;
  }
}

long kkt_oracle_model_offset(const kkt_oracle *o, int block, int index) {
  if (block < 0 || block >= KKT_NUM_BLOCKS || index < 0 ||
      index >= (block <= KKT_NODE_DG_DX ? o->N : o->E))
    return -1;
  return o->off[block][index];
}

long kkt_oracle_vector_offset(const kkt_oracle *o, int table, int index) {
  if (table < 0 || table > KKT_Z_EDGE || index < 0 || index >= o->N)
    return -1;
  return o->voff[table][index];
}

const double *kkt_oracle_lqr_block(const kkt_oracle *o, char name, int index) {
  switch (name) {
  case 'Q': return o->Q_mod[index];
  case 'M': return o->M_mod[index];
  case 'R': return o->R_mod[index];
  case 'q': return o->q_mod[index];
  case 'r': return o->r_mod[index];
  case 'c': return o->c_mod[index];
  case 'd': return o->dyn_r2[index];
  default: return NULL;
  }
}

/* add_weighted_state_jacobian_product, helpers.cpp:116-136: lower triangle of
 * Q += J^T diag(weights) J, J (rows x n) column-major. */
static void add_weighted_state(double *Q, int n, const double *J, int rows, const double *weights) {
  for (int k = 0; k < rows; ++k) {
    const double weight = weights[k];
    for (int col = 0; col < n; ++col) {
      const double wj = weight * J[k + (long)rows * col];
      if (wj == 0.0)
        continue;
      for (int row = col; row < n; ++row) {
        const double jx = J[k + (long)rows * row];
        if (jx == 0.0)
          continue;
        Q[row + (long)n * col] += wj * jx;
      }
    }
  }
}

/* add_weighted_control_jacobian_products, helpers.cpp:79-114 */
static void add_weighted_control(double *M, double *R, int n, int m, const double *Jx,
                                 const double *Ju, int rows, const double *weights) {
  for (int k = 0; k < rows; ++k) {
    const double weight = weights[k];
    for (int col = 0; col < m; ++col) {
      const double wj = weight * Ju[k + (long)rows * col];
      if (wj == 0.0)
        continue;
      for (int row = 0; row < n; ++row) {
        const double jx = Jx[k + (long)rows * row];
        if (jx == 0.0)
          continue;
        M[row + (long)n * col] += jx * wj;
      }
    }
    for (int col = 0; col < m; ++col) {
      const double wj = weight * Ju[k + (long)rows * col];
      if (wj == 0.0)
        continue;
      for (int row = col; row < m; ++row) {
        const double ju = Ju[k + (long)rows * row];
        if (ju == 0.0)
          continue;
        R[row + (long)m * col] += wj * ju;
      }
    }
  }
}

/* subtract_weighted_jacobian_rhs, helpers.cpp:138-153 */
static void subtract_weighted_rhs(double *result, int cols, const double *J, int rows,
                                  const double *weights, const double *rhs) {
  for (int k = 0; k < rows; ++k) {
    const double wr = weights[k] * rhs[k];
    for (int col = 0; col < cols; ++col) {
      const double j = J[k + (long)rows * col];
      if (j == 0.0)
        continue;
      result[col] -= j * wr;
    }
  }
}

/* mirror_lower_to_upper, helpers.cpp:155-158 */
static void mirror(double *A, int n) {
  for (int col = 0; col < n; ++col)
    for (int row = col + 1; row < n; ++row)
      A[col + (long)n * row] = A[row + (long)n * col];
}

#define BLK(o, model, block, index) ((model) + (o)->off[block][index])

/* CallbackProvider::factor, helpers.cpp:242-370 (theta_dim == 0) */
int kkt_oracle_factor(kkt_oracle *o, const double *model, const double *w,
                      const double *r1, const double *r2, const double *r3) {
  if (!o->input_valid)
    return KKT_ORACLE_INVALID_INPUT; /* :244-246 */
  const int E = o->E, N = o->N;
  for (int i = 0; i < N; ++i) { /* :251-278 */
    for (int row = 0; row < o->sd[i]; ++row) {
      const double reg = r2[o->voff[KKT_Y_DYN][i] + row];
      if (reg <= 0.0)
        return KKT_ORACLE_NONPOSITIVE_REGULARIZATION;
      o->dyn_r2[i][row] = reg;
    }
    for (int row = 0; row < o->ncd[i]; ++row) {
      const double reg = r2[o->voff[KKT_Y_NODE_C][i] + row];
      if (reg <= 0.0)
        return KKT_ORACLE_NONPOSITIVE_REGULARIZATION;
      o->node_c_r2_inv[i][row] = 1.0 / reg;
    }
    for (int row = 0; row < o->ngd[i]; ++row) {
      const int at = o->voff[KKT_Z_NODE][i] + row;
      const double reg = w[at] + r3[at];
      if (reg <= 0.0)
        return KKT_ORACLE_NONPOSITIVE_REGULARIZATION;
      o->node_mod_w_inv[i][row] = 1.0 / reg;
    }
  }
  for (int e = 0; e < E; ++e) { /* :280-297 */
    for (int row = 0; row < o->ecd[e]; ++row) {
      const double reg = r2[o->voff[KKT_Y_EDGE_C][e] + row];
      if (reg <= 0.0)
        return KKT_ORACLE_NONPOSITIVE_REGULARIZATION;
      o->edge_c_r2_inv[e][row] = 1.0 / reg;
    }
    for (int row = 0; row < o->egd[e]; ++row) {
      const int at = o->voff[KKT_Z_EDGE][e] + row;
      const double reg = w[at] + r3[at];
      if (reg <= 0.0)
        return KKT_ORACLE_NONPOSITIVE_REGULARIZATION;
      o->edge_mod_w_inv[e][row] = 1.0 / reg;
    }
  }

  for (int i = 0; i < N; ++i) { /* :299-318 */
    const int n = o->sd[i];
    const double *Q = BLK(o, model, KKT_NODE_D2L_DX2, i);
    double *Qm = o->Q_mod[i];
    for (int col = 0; col < n; ++col)
      for (int row = col; row < n; ++row)
        Qm[row + (long)n * col] = Q[row + (long)n * col];
    for (int d = 0; d < n; ++d)
      Qm[d + (long)n * d] += r1[o->voff[KKT_X_STATE][i] + d];
    add_weighted_state(Qm, n, BLK(o, model, KKT_NODE_DC_DX, i), o->ncd[i], o->node_c_r2_inv[i]);
    add_weighted_state(Qm, n, BLK(o, model, KKT_NODE_DG_DX, i), o->ngd[i], o->node_mod_w_inv[i]);
  }

  for (int e = 0; e < E; ++e) { /* :320-355 */
    const int parent = o->ws.edge_parents[e];
    const int n = o->sd[parent], m = o->cd[e], c = o->ecd[e], g = o->egd[e];
    const double *eQ = BLK(o, model, KKT_EDGE_D2L_DX2, e);
    const double *Jxc = BLK(o, model, KKT_EDGE_DC_DX, e), *Juc = BLK(o, model, KKT_EDGE_DC_DU, e);
    const double *Jxg = BLK(o, model, KKT_EDGE_DG_DX, e), *Jug = BLK(o, model, KKT_EDGE_DG_DU, e);
    double *Qm = o->Q_mod[parent];
    for (int col = 0; col < n; ++col)
      for (int row = col; row < n; ++row)
        Qm[row + (long)n * col] += eQ[row + (long)n * col];
    add_weighted_state(Qm, n, Jxc, c, o->edge_c_r2_inv[e]);
    add_weighted_state(Qm, n, Jxg, g, o->edge_mod_w_inv[e]);

    double *Mm = o->M_mod[e], *Rm = o->R_mod[e];
    const double *M = BLK(o, model, KKT_EDGE_D2L_DXDU, e), *R = BLK(o, model, KKT_EDGE_D2L_DU2, e);
    memcpy(Mm, M, sizeof(double) * (size_t)n * (size_t)m);
    for (int col = 0; col < m; ++col)
      for (int row = col; row < m; ++row)
        Rm[row + (long)m * col] = R[row + (long)m * col];
    for (int d = 0; d < m; ++d)
      Rm[d + (long)m * d] += r1[o->voff[KKT_X_CONTROL][e] + d];
    add_weighted_control(Mm, Rm, n, m, Jxc, Juc, c, o->edge_c_r2_inv[e]);
    add_weighted_control(Mm, Rm, n, m, Jxg, Jug, g, o->edge_mod_w_inv[e]);
    mirror(Rm, m);
    o->A[e] = (double *)BLK(o, model, KKT_EDGE_DDYN_DX, e);
    o->B[e] = (double *)BLK(o, model, KKT_EDGE_DDYN_DU, e);
  }
  for (int i = 0; i < N; ++i) /* :357-361 */
    mirror(o->Q_mod[i], o->sd[i]);

  lqr_oracle_problem *p = &o->prob; /* :363-370 */
  p->Q = o->Q_mod, p->M = o->M_mod, p->R = o->R_mod;
  p->A = o->A, p->B = o->B, p->delta = o->dyn_r2;
  return lqr_oracle_factor(p, &o->ws);
}

/* CallbackProvider::solve -> solve_stagewise_kkt_matrix(b, sol, 1),
 * helpers.cpp:749-893 */
void kkt_oracle_solve(kkt_oracle *o, const double *model, const double *b, double *sol) {
  const int E = o->E, N = o->N, x_dim = o->x_dim, y_dim = o->y_dim;
  for (int i = 0; i < N; ++i) { /* :752-779 */
    const int n = o->sd[i];
    double *q = o->q_mod[i], *cm = o->c_mod[i];
    for (int d = 0; d < n; ++d)
      q[d] = -b[o->voff[KKT_X_STATE][i] + d];
    subtract_weighted_rhs(q, n, BLK(o, model, KKT_NODE_DC_DX, i), o->ncd[i], o->node_c_r2_inv[i],
                          b + x_dim + o->voff[KKT_Y_NODE_C][i]);
    subtract_weighted_rhs(q, n, BLK(o, model, KKT_NODE_DG_DX, i), o->ngd[i], o->node_mod_w_inv[i],
                          b + x_dim + y_dim + o->voff[KKT_Z_NODE][i]);
    for (int d = 0; d < n; ++d)
      cm[d] = -b[x_dim + o->voff[KKT_Y_DYN][i] + d];
  }
  for (int e = 0; e < E; ++e) { /* :781-812 */
    const int parent = o->ws.edge_parents[e];
    const int n = o->sd[parent], m = o->cd[e], c = o->ecd[e], g = o->egd[e];
    const double *byc = b + x_dim + o->voff[KKT_Y_EDGE_C][e];
    const double *bz = b + x_dim + y_dim + o->voff[KKT_Z_EDGE][e];
    double *qp = o->q_mod[parent], *r = o->r_mod[e];
    subtract_weighted_rhs(qp, n, BLK(o, model, KKT_EDGE_DC_DX, e), c, o->edge_c_r2_inv[e], byc);
    subtract_weighted_rhs(qp, n, BLK(o, model, KKT_EDGE_DG_DX, e), g, o->edge_mod_w_inv[e], bz);
    for (int d = 0; d < m; ++d)
      r[d] = -b[o->voff[KKT_X_CONTROL][e] + d];
    subtract_weighted_rhs(r, m, BLK(o, model, KKT_EDGE_DC_DU, e), c, o->edge_c_r2_inv[e], byc);
    subtract_weighted_rhs(r, m, BLK(o, model, KKT_EDGE_DG_DU, e), g, o->edge_mod_w_inv[e], bz);
    o->A[e] = (double *)BLK(o, model, KKT_EDGE_DDYN_DX, e);
    o->B[e] = (double *)BLK(o, model, KKT_EDGE_DDYN_DU, e);
  }

  lqr_oracle_problem *p = &o->prob; /* :814-826: outputs alias the flat solution */
  p->q = o->q_mod, p->r = o->r_mod, p->c = o->c_mod;
  for (int i = 0; i < N; ++i) {
    o->x[i] = sol + o->voff[KKT_X_STATE][i];
    o->y[i] = sol + x_dim + o->voff[KKT_Y_DYN][i];
  }
  for (int e = 0; e < E; ++e)
    o->u[e] = sol + o->voff[KKT_X_CONTROL][e];
  lqr_oracle_solve(p, &o->ws, o->x, o->u, o->y);

  for (int i = 0; i < N; ++i) { /* :828-855: multipliers of node constraints */
    const int n = o->sd[i], c = o->ncd[i], g = o->ngd[i];
    const double *x = sol + o->voff[KKT_X_STATE][i];
    const double *Jc = BLK(o, model, KKT_NODE_DC_DX, i), *Jg = BLK(o, model, KKT_NODE_DG_DX, i);
    double *yc = sol + x_dim + o->voff[KKT_Y_NODE_C][i];
    double *z = sol + x_dim + y_dim + o->voff[KKT_Z_NODE][i];
    for (int k = 0; k < c; ++k) {
      double acc = 0.0;
      for (int col = 0; col < n; ++col)
        acc += Jc[k + (long)c * col] * x[col];
      yc[k] = (acc - b[x_dim + o->voff[KKT_Y_NODE_C][i] + k]) * o->node_c_r2_inv[i][k];
    }
    for (int k = 0; k < g; ++k) {
      double acc = 0.0;
      for (int col = 0; col < n; ++col)
        acc += Jg[k + (long)g * col] * x[col];
      z[k] = (acc - b[x_dim + y_dim + o->voff[KKT_Z_NODE][i] + k]) * o->node_mod_w_inv[i][k];
    }
  }
  for (int e = 0; e < E; ++e) { /* :857-892: multipliers of edge constraints */
    const int parent = o->ws.edge_parents[e];
    const int n = o->sd[parent], m = o->cd[e], c = o->ecd[e], g = o->egd[e];
    const double *x = sol + o->voff[KKT_X_STATE][parent], *u = sol + o->voff[KKT_X_CONTROL][e];
    const double *Jxc = BLK(o, model, KKT_EDGE_DC_DX, e), *Juc = BLK(o, model, KKT_EDGE_DC_DU, e);
    const double *Jxg = BLK(o, model, KKT_EDGE_DG_DX, e), *Jug = BLK(o, model, KKT_EDGE_DG_DU, e);
    double *yc = sol + x_dim + o->voff[KKT_Y_EDGE_C][e];
    double *z = sol + x_dim + y_dim + o->voff[KKT_Z_EDGE][e];
    for (int k = 0; k < c; ++k) {
      double acc = 0.0, acu = 0.0;
      for (int col = 0; col < n; ++col)
        acc += Jxc[k + (long)c * col] * x[col];
      for (int col = 0; col < m; ++col)
        acu += Juc[k + (long)c * col] * u[col];
      yc[k] = ((acc + acu) - b[x_dim + o->voff[KKT_Y_EDGE_C][e] + k]) * o->edge_c_r2_inv[e][k];
    }
    for (int k = 0; k < g; ++k) {
      double acc = 0.0, acu = 0.0;
      for (int col = 0; col < n; ++col)
        acc += Jxg[k + (long)g * col] * x[col];
      for (int col = 0; col < m; ++col)
        acu += Jug[k + (long)g * col] * u[col];
      z[k] = ((acc + acu) - b[x_dim + y_dim + o->voff[KKT_Z_EDGE][e] + k]) * o->edge_mod_w_inv[e][k];
    }
  }
}

/* y (rows) += J x, J (rows x cols) column-major */
static void gemv_n(double *y, const double *J, int rows, int cols, const double *x, double sign) {
  for (int col = 0; col < cols; ++col) {
    const double xs = sign * x[col];
    for (int row = 0; row < rows; ++row)
      y[row] += J[row + (long)rows * col] * xs;
  }
}
/* y (cols) += J^T x */
static void gemv_t(double *y, const double *J, int rows, int cols, const double *x) {
  for (int col = 0; col < cols; ++col) {
    double acc = 0.0;
    for (int row = 0; row < rows; ++row)
      acc += J[row + (long)rows * col] * x[row];
    y[col] += acc;
  }
}

/* CallbackProvider::add_Kx_to_y, helpers.cpp:953-976, with add_Hx_to_y
 * (:979-1022), add_Cx_to_y (:1070-1126), add_CTx_to_y (:1161-1219),
 * add_Gx_to_y (:1252-1283), add_GTx_to_y (:1311-1342); theta_dim == 0. */
void kkt_oracle_add_Kx_to_y(const kkt_oracle *o, const double *model, const double *w,
                            const double *r1, const double *r2, const double *r3,
                            const double *xv, double *yv) {
  const int E = o->E, N = o->N, x_dim = o->x_dim, y_dim = o->y_dim, z_dim = o->z_dim;
  const double *x_x = xv, *x_y = xv + x_dim, *x_z = xv + x_dim + y_dim;
  double *y_x = yv, *y_y = yv + x_dim, *y_z = yv + x_dim + y_dim;
  const int root = o->ws.preorder_nodes[0];

  /* Hx */
  for (int i = 0; i < N; ++i)
    gemv_n(y_x + o->voff[KKT_X_STATE][i], BLK(o, model, KKT_NODE_D2L_DX2, i), o->sd[i], o->sd[i],
           x_x + o->voff[KKT_X_STATE][i], 1.0);
  for (int e = 0; e < E; ++e) {
    const int parent = o->ws.edge_parents[e], n = o->sd[parent], m = o->cd[e];
    const double *xp = x_x + o->voff[KKT_X_STATE][parent], *ue = x_x + o->voff[KKT_X_CONTROL][e];
    double *yp = y_x + o->voff[KKT_X_STATE][parent], *yu = y_x + o->voff[KKT_X_CONTROL][e];
    gemv_n(yp, BLK(o, model, KKT_EDGE_D2L_DX2, e), n, n, xp, 1.0);
    gemv_n(yp, BLK(o, model, KKT_EDGE_D2L_DXDU, e), n, m, ue, 1.0);
    gemv_t(yu, BLK(o, model, KKT_EDGE_D2L_DXDU, e), n, m, xp);
    gemv_n(yu, BLK(o, model, KKT_EDGE_D2L_DU2, e), m, m, ue, 1.0);
  }
  /* Cx: root row is -x_root (the initial-state constraint) */
  for (int d = 0; d < o->sd[root]; ++d)
    y_y[o->voff[KKT_Y_DYN][root] + d] -= x_x[o->voff[KKT_X_STATE][root] + d];
  for (int i = 0; i < N; ++i)
    gemv_n(y_y + o->voff[KKT_Y_NODE_C][i], BLK(o, model, KKT_NODE_DC_DX, i), o->ncd[i], o->sd[i],
           x_x + o->voff[KKT_X_STATE][i], 1.0);
  for (int e = 0; e < E; ++e) {
    const int parent = o->ws.edge_parents[e], child = o->ws.edge_children[e];
    const int np = o->sd[parent], nc = o->sd[child], m = o->cd[e], c = o->ecd[e];
    const double *xp = x_x + o->voff[KKT_X_STATE][parent], *ue = x_x + o->voff[KKT_X_CONTROL][e];
    double *yd = y_y + o->voff[KKT_Y_DYN][child], *yc = y_y + o->voff[KKT_Y_EDGE_C][e];
    gemv_n(yd, BLK(o, model, KKT_EDGE_DDYN_DX, e), nc, np, xp, 1.0);
    gemv_n(yd, BLK(o, model, KKT_EDGE_DDYN_DU, e), nc, m, ue, 1.0);
    for (int d = 0; d < nc; ++d)
      yd[d] -= x_x[o->voff[KKT_X_STATE][child] + d];
    gemv_n(yc, BLK(o, model, KKT_EDGE_DC_DX, e), c, np, xp, 1.0);
    gemv_n(yc, BLK(o, model, KKT_EDGE_DC_DU, e), c, m, ue, 1.0);
  }
  /* C^T y */
  for (int d = 0; d < o->sd[root]; ++d)
    y_x[o->voff[KKT_X_STATE][root] + d] -= x_y[o->voff[KKT_Y_DYN][root] + d];
  for (int i = 0; i < N; ++i)
    gemv_t(y_x + o->voff[KKT_X_STATE][i], BLK(o, model, KKT_NODE_DC_DX, i), o->ncd[i], o->sd[i],
           x_y + o->voff[KKT_Y_NODE_C][i]);
  for (int e = 0; e < E; ++e) {
    const int parent = o->ws.edge_parents[e], child = o->ws.edge_children[e];
    const int np = o->sd[parent], nc = o->sd[child], m = o->cd[e], c = o->ecd[e];
    const double *dyn = x_y + o->voff[KKT_Y_DYN][child], *cv = x_y + o->voff[KKT_Y_EDGE_C][e];
    double *yp = y_x + o->voff[KKT_X_STATE][parent], *ych = y_x + o->voff[KKT_X_STATE][child];
    double *yu = y_x + o->voff[KKT_X_CONTROL][e];
    gemv_t(yp, BLK(o, model, KKT_EDGE_DDYN_DX, e), nc, np, dyn);
    gemv_t(yp, BLK(o, model, KKT_EDGE_DC_DX, e), c, np, cv);
    for (int d = 0; d < nc; ++d)
      ych[d] -= dyn[d];
    gemv_t(yu, BLK(o, model, KKT_EDGE_DDYN_DU, e), nc, m, dyn);
    gemv_t(yu, BLK(o, model, KKT_EDGE_DC_DU, e), c, m, cv);
  }
  /* Gx, G^T z */
  for (int i = 0; i < N; ++i) {
    gemv_n(y_z + o->voff[KKT_Z_NODE][i], BLK(o, model, KKT_NODE_DG_DX, i), o->ngd[i], o->sd[i],
           x_x + o->voff[KKT_X_STATE][i], 1.0);
    gemv_t(y_x + o->voff[KKT_X_STATE][i], BLK(o, model, KKT_NODE_DG_DX, i), o->ngd[i], o->sd[i],
           x_z + o->voff[KKT_Z_NODE][i]);
  }
  for (int e = 0; e < E; ++e) {
    const int parent = o->ws.edge_parents[e], n = o->sd[parent], m = o->cd[e], g = o->egd[e];
    const double *xp = x_x + o->voff[KKT_X_STATE][parent], *ue = x_x + o->voff[KKT_X_CONTROL][e];
    const double *ze = x_z + o->voff[KKT_Z_EDGE][e];
    gemv_n(y_z + o->voff[KKT_Z_EDGE][e], BLK(o, model, KKT_EDGE_DG_DX, e), g, n, xp, 1.0);
    gemv_n(y_z + o->voff[KKT_Z_EDGE][e], BLK(o, model, KKT_EDGE_DG_DU, e), g, m, ue, 1.0);
    gemv_t(y_x + o->voff[KKT_X_STATE][parent], BLK(o, model, KKT_EDGE_DG_DX, e), g, n, ze);
    gemv_t(y_x + o->voff[KKT_X_CONTROL][e], BLK(o, model, KKT_EDGE_DG_DU, e), g, m, ze);
  }
  /* regularization diagonal, :965-975 */
  for (int i = 0; i < x_dim; ++i)
    y_x[i] += r1[i] * x_x[i];
  for (int i = 0; i < y_dim; ++i)
    y_y[i] -= r2[i] * x_y[i];
  for (int i = 0; i < z_dim; ++i)
    y_z[i] -= (w[i] + r3[i]) * x_z[i];
}

int kkt_oracle_batch(const kkt_oracle *o, long batch, const double *model,
                     const double *w, const double *r1, const double *r2,
                     const double *r3, const double *b, double *sol,
                     int *status, int threads) {
  const long kkt = (long)o->x_dim + o->y_dim + o->z_dim;
#ifdef _OPENMP
  if (threads < 1)
    threads = 1;
#pragma omp parallel num_threads(threads)
#endif
  {
    kkt_oracle *mine = kkt_oracle_create(o->E, o->root, o->parents, o->children, o->sd, o->cd,
                                         o->ncd, o->ngd, o->ecd, o->egd);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long p = 0; p < batch; ++p) {
      const double *mp = model + p * o->model_len;
      status[p] = kkt_oracle_factor(mine, mp, w + p * o->z_dim, r1 + p * o->x_dim,
                                    r2 + p * o->y_dim, r3 + p * o->z_dim);
      if (status[p] == KKT_ORACLE_SUCCESS)
        kkt_oracle_solve(mine, mp, b + p * kkt, sol + p * kkt);
    }
    kkt_oracle_destroy(mine);
  }
  (void)threads;
  return 0;
}


/* ------------------------------------------------------------------------
 * theta_dim > 0
 * ------------------------------------------------------------------------ */
void kkt_oracle_set_theta(kkt_oracle *o, int theta_dim) {
  if (!o->input_valid || theta_dim <= 0)
    return;
  const int E = o->E, N = o->N, p = theta_dim;
  o->p = p;
  for (int b = 0; b < KKT_TH_NUM_BLOCKS; ++b)
    o->toff[b] = (long *)calloc((size_t)N, sizeof(long));
  long at = 0;
  for (int i = 0; i < N; ++i) {
    o->toff[KKT_TH_NODE_DXDTH][i] = at, at += (long)o->sd[i] * p;
    o->toff[KKT_TH_NODE_DC][i] = at, at += (long)o->ncd[i] * p;
    o->toff[KKT_TH_NODE_DG][i] = at, at += (long)o->ngd[i] * p;
    o->toff[KKT_TH_NODE_DTH2][i] = at, at += (long)p * p;
    if (i < E) {
      const int e = i;
      o->toff[KKT_TH_EDGE_DXDTH][e] = at, at += (long)o->sd[o->parents[e]] * p;
      o->toff[KKT_TH_EDGE_DUDTH][e] = at, at += (long)o->cd[e] * p;
      o->toff[KKT_TH_EDGE_DDYN][e] = at, at += (long)o->sd[o->children[e]] * p;
      o->toff[KKT_TH_EDGE_DC][e] = at, at += (long)o->ecd[e] * p;
      o->toff[KKT_TH_EDGE_DG][e] = at, at += (long)o->egd[e] * p;
      o->toff[KKT_TH_EDGE_DTH2][e] = at, at += (long)p * p;
    }
  }
  o->theta_len = at;
  const long skkt = (long)o->x_dim + o->y_dim + o->z_dim;
  o->J_theta = (double *)calloc((size_t)(skkt * p + 1), sizeof(double));
  o->K_inv_J_theta = (double *)calloc((size_t)(skkt * p + 1), sizeof(double));
  o->S_factor = (double *)calloc((size_t)p * p, sizeof(double));
  o->theta_rhs = (double *)calloc((size_t)p, sizeof(double));
  o->stagewise_rhs = (double *)calloc((size_t)skkt + 1, sizeof(double));
  o->stagewise_sol = (double *)calloc((size_t)skkt + 1, sizeof(double));
}

long kkt_oracle_theta_len(const kkt_oracle *o) { return o->theta_len; }

long kkt_oracle_theta_offset(const kkt_oracle *o, int block, int index) {
  if (o->p <= 0 || block < 0 || block >= KKT_TH_NUM_BLOCKS || index < 0 ||
      index >= (block <= KKT_TH_NODE_DTH2 ? o->N : o->E))
    return -1;
  return o->toff[block][index];
}

#define TBLK(o, tm, block, index) ((tm) + (o)->toff[block][index])

/* dst (rows x p block at row offset `at` of the skkt x p matrix J) (+)= src (rows x p) */
static void put_rows(double *J, long skkt, int at, const double *src, int rows, int p, int accumulate) {
  for (int col = 0; col < p; ++col)
    for (int r = 0; r < rows; ++r) {
      double *dst = &J[at + r + skkt * col];
      *dst = (accumulate ? *dst : 0.0) + src[r + (long)rows * col];
    }
}

/* form_theta_jacobian, helpers.cpp:190-240 */
static void form_theta_jacobian(kkt_oracle *o, const double *tm) {
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim, yd = o->y_dim;
  const long skkt = (long)sx + yd + o->z_dim;
  memset(o->J_theta, 0, sizeof(double) * (size_t)(skkt * p));
  for (int i = 0; i < N; ++i) {
    put_rows(o->J_theta, skkt, o->voff[KKT_X_STATE][i], TBLK(o, tm, KKT_TH_NODE_DXDTH, i), o->sd[i], p, 0);
    put_rows(o->J_theta, skkt, sx + o->voff[KKT_Y_NODE_C][i], TBLK(o, tm, KKT_TH_NODE_DC, i), o->ncd[i], p, 0);
    put_rows(o->J_theta, skkt, sx + yd + o->voff[KKT_Z_NODE][i], TBLK(o, tm, KKT_TH_NODE_DG, i), o->ngd[i], p, 0);
  }
  for (int e = 0; e < E; ++e) {
    const int parent = o->ws.edge_parents[e], child = o->ws.edge_children[e];
    put_rows(o->J_theta, skkt, o->voff[KKT_X_STATE][parent], TBLK(o, tm, KKT_TH_EDGE_DXDTH, e), o->sd[parent], p, 1);
    put_rows(o->J_theta, skkt, o->voff[KKT_X_CONTROL][e], TBLK(o, tm, KKT_TH_EDGE_DUDTH, e), o->cd[e], p, 0);
    put_rows(o->J_theta, skkt, sx + o->voff[KKT_Y_DYN][child], TBLK(o, tm, KKT_TH_EDGE_DDYN, e), o->sd[child], p, 0);
    put_rows(o->J_theta, skkt, sx + o->voff[KKT_Y_EDGE_C][e], TBLK(o, tm, KKT_TH_EDGE_DC, e), o->ecd[e], p, 0);
    put_rows(o->J_theta, skkt, sx + yd + o->voff[KKT_Z_EDGE][e], TBLK(o, tm, KKT_TH_EDGE_DG, e), o->egd[e], p, 0);
  }
}

/* CallbackProvider::factor with theta_dim > 0, helpers.cpp:242-407 */
int kkt_oracle_factor_theta(kkt_oracle *o, const double *model, const double *tm, const double *w,
                            const double *r1, const double *r2, const double *r3) {
  const int status = kkt_oracle_factor(o, model, w, r1, r2, r3);
  if (status != KKT_ORACLE_SUCCESS || o->p <= 0)
    return status;
  const int p = o->p, sx = o->x_dim;
  const long skkt = (long)sx + o->y_dim + o->z_dim;
  form_theta_jacobian(o, tm); /* :376 */
  for (int col = 0; col < p; ++col) /* solve_stagewise_kkt_matrix(J_theta, K_inv_J_theta, p), :387 */
    kkt_oracle_solve(o, model, o->J_theta + skkt * col, o->K_inv_J_theta + skkt * col);
  double *S = o->S_factor; /* :389-402 */
  memset(S, 0, sizeof(double) * (size_t)p * p);
  for (int i = 0; i < o->N; ++i)
    for (int k = 0; k < p * p; ++k)
      S[k] += TBLK(o, tm, KKT_TH_NODE_DTH2, i)[k];
  for (int e = 0; e < o->E; ++e)
    for (int k = 0; k < p * p; ++k)
      S[k] += TBLK(o, tm, KKT_TH_EDGE_DTH2, e)[k];
  for (int d = 0; d < p; ++d)
    S[d + (long)p * d] += r1[sx + d];
  for (int col = 0; col < p; ++col)
    for (int row = 0; row < p; ++row) {
      double acc = 0.0;
      for (long r = 0; r < skkt; ++r)
        acc += o->J_theta[r + skkt * row] * o->K_inv_J_theta[r + skkt * col];
      S[row + (long)p * col] -= acc;
    }
  /* Eigen::LLT in place, lower (:403-406): pivot <= 0 -> NumericalIssue */
  for (int k = 0; k < p; ++k) {
    double d = S[k + (long)p * k];
    for (int j = 0; j < k; ++j)
      d -= S[k + (long)p * j] * S[k + (long)p * j];
    if (!(d > 0.0))
      return KKT_ORACLE_THETA_SCHUR_FAILURE;
    d = sqrt(d);
    S[k + (long)p * k] = d;
    for (int i = k + 1; i < p; ++i) {
      double v = S[i + (long)p * k];
      for (int j = 0; j < k; ++j)
        v -= S[i + (long)p * j] * S[k + (long)p * j];
      S[i + (long)p * k] = v / d;
    }
  }
  return KKT_ORACLE_SUCCESS;
}

/* CallbackProvider::solve with theta_dim > 0, helpers.cpp:896-951 */
void kkt_oracle_solve_theta(kkt_oracle *o, const double *model, const double *tm, const double *b,
                            double *sol) {
  (void)tm;
  if (o->p <= 0) {
    kkt_oracle_solve(o, model, b, sol);
    return;
  }
  const int p = o->p, sx = o->x_dim, yd = o->y_dim, zd = o->z_dim;
  const long skkt = (long)sx + yd + zd;
  const double *b_theta = b + sx, *b_y = b + sx + p, *b_z = b_y + yd;
  memcpy(o->stagewise_rhs, b, sizeof(double) * (size_t)sx); /* :911-916 */
  memcpy(o->stagewise_rhs + sx, b_y, sizeof(double) * (size_t)yd);
  memcpy(o->stagewise_rhs + sx + yd, b_z, sizeof(double) * (size_t)zd);
  kkt_oracle_solve(o, model, o->stagewise_rhs, o->stagewise_sol); /* :918 */
  for (int a = 0; a < p; ++a) { /* theta_rhs = b_theta - J_theta^T K^-1 b, :920-926 */
    double acc = 0.0;
    for (long r = 0; r < skkt; ++r)
      acc += o->J_theta[r + skkt * a] * o->stagewise_sol[r];
    o->theta_rhs[a] = b_theta[a] - acc;
  }
  const double *L = o->S_factor; /* two triangular solves, :928-934 */
  for (int i = 0; i < p; ++i) {
    double v = o->theta_rhs[i];
    for (int j = 0; j < i; ++j)
      v -= L[i + (long)p * j] * o->theta_rhs[j];
    o->theta_rhs[i] = v / L[i + (long)p * i];
  }
  for (int i = p - 1; i >= 0; --i) {
    double v = o->theta_rhs[i];
    for (int j = i + 1; j < p; ++j)
      v -= L[j + (long)p * i] * o->theta_rhs[j];
    o->theta_rhs[i] = v / L[i + (long)p * i];
  }
  for (long r = 0; r < skkt; ++r) { /* stagewise_solution -= K^-1 J_theta theta, :936-939 */
    double acc = 0.0;
    for (int a = 0; a < p; ++a)
      acc += o->K_inv_J_theta[r + skkt * a] * o->theta_rhs[a];
    o->stagewise_sol[r] -= acc;
  }
  memcpy(sol, o->stagewise_sol, sizeof(double) * (size_t)sx); /* :941-950 */
  memcpy(sol + sx, o->theta_rhs, sizeof(double) * (size_t)p);
  memcpy(sol + sx + p, o->stagewise_sol + sx, sizeof(double) * (size_t)(yd + zd));
}

/* theta terms of add_Hx/Cx/CTx/Gx/GTx_to_y (helpers.cpp:1023-1066, 1128-1158,
 * 1221-1249, 1285-1308, 1344-1367) on top of the stagewise operator */
void kkt_oracle_add_Kx_to_y_theta(const kkt_oracle *o, const double *model, const double *tm,
                                  const double *w, const double *r1, const double *r2, const double *r3,
                                  const double *xv, double *yv) {
  if (o->p <= 0) {
    kkt_oracle_add_Kx_to_y(o, model, w, r1, r2, r3, xv, yv);
    return;
  }
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim, yd = o->y_dim, zd = o->z_dim;
  const long skkt = (long)sx + yd + zd;
  /* stagewise part on compacted copies ([x | y | z] without theta) */
  double *xs = (double *)calloc((size_t)skkt + 1, sizeof(double));
  double *ys = (double *)calloc((size_t)skkt + 1, sizeof(double));
  double *r1s = (double *)calloc((size_t)sx + 1, sizeof(double));
  memcpy(xs, xv, sizeof(double) * (size_t)sx);
  memcpy(xs + sx, xv + sx + p, sizeof(double) * (size_t)(yd + zd));
  memcpy(r1s, r1, sizeof(double) * (size_t)sx);
  kkt_oracle_add_Kx_to_y(o, model, w, r1s, r2, r3, xs, ys);
  for (int k = 0; k < sx; ++k)
    yv[k] += ys[k];
  for (int k = 0; k < yd + zd; ++k)
    yv[sx + p + k] += ys[sx + k];
  free(xs), free(ys), free(r1s);

  const double *x_x = xv, *theta = xv + sx, *x_y = xv + sx + p, *x_z = x_y + yd;
  double *y_x = yv, *y_theta = yv + sx, *y_y = yv + sx + p, *y_z = y_y + yd;
  for (int i = 0; i < N; ++i) {
    const int n = o->sd[i], c = o->ncd[i], g = o->ngd[i];
    const double *Hxt = TBLK(o, tm, KKT_TH_NODE_DXDTH, i);
    gemv_n(y_x + o->voff[KKT_X_STATE][i], Hxt, n, p, theta, 1.0);        /* :1036 */
    gemv_t(y_theta, Hxt, n, p, x_x + o->voff[KKT_X_STATE][i]);            /* :1037 */
    gemv_n(y_theta, TBLK(o, tm, KKT_TH_NODE_DTH2, i), p, p, theta, 1.0);  /* :1038-1040 */
    gemv_n(y_y + o->voff[KKT_Y_NODE_C][i], TBLK(o, tm, KKT_TH_NODE_DC, i), c, p, theta, 1.0);
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_NODE_DC, i), c, p, x_y + o->voff[KKT_Y_NODE_C][i]);
    gemv_n(y_z + o->voff[KKT_Z_NODE][i], TBLK(o, tm, KKT_TH_NODE_DG, i), g, p, theta, 1.0);
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_NODE_DG, i), g, p, x_z + o->voff[KKT_Z_NODE][i]);
  }
  for (int e = 0; e < E; ++e) {
    const int parent = o->ws.edge_parents[e], child = o->ws.edge_children[e];
    const int n = o->sd[parent], nc = o->sd[child], m = o->cd[e], c = o->ecd[e], g = o->egd[e];
    const double *Hxt = TBLK(o, tm, KKT_TH_EDGE_DXDTH, e), *Hut = TBLK(o, tm, KKT_TH_EDGE_DUDTH, e);
    gemv_n(y_x + o->voff[KKT_X_STATE][parent], Hxt, n, p, theta, 1.0);
    gemv_n(y_x + o->voff[KKT_X_CONTROL][e], Hut, m, p, theta, 1.0);
    gemv_t(y_theta, Hxt, n, p, x_x + o->voff[KKT_X_STATE][parent]);
    gemv_t(y_theta, Hut, m, p, x_x + o->voff[KKT_X_CONTROL][e]);
    gemv_n(y_theta, TBLK(o, tm, KKT_TH_EDGE_DTH2, e), p, p, theta, 1.0);
    gemv_n(y_y + o->voff[KKT_Y_DYN][child], TBLK(o, tm, KKT_TH_EDGE_DDYN, e), nc, p, theta, 1.0);
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_EDGE_DDYN, e), nc, p, x_y + o->voff[KKT_Y_DYN][child]);
    gemv_n(y_y + o->voff[KKT_Y_EDGE_C][e], TBLK(o, tm, KKT_TH_EDGE_DC, e), c, p, theta, 1.0);
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_EDGE_DC, e), c, p, x_y + o->voff[KKT_Y_EDGE_C][e]);
    gemv_n(y_z + o->voff[KKT_Z_EDGE][e], TBLK(o, tm, KKT_TH_EDGE_DG, e), g, p, theta, 1.0);
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_EDGE_DG, e), g, p, x_z + o->voff[KKT_Z_EDGE][e]);
  }
  for (int d = 0; d < p; ++d) /* r1 on the theta block, :965-967 */
    y_theta[d] += r1[sx + d] * theta[d];
}

/* ------------------------------------------------------------------------
 * The five block operators CallbackProvider hands to SIP one by one
 * (sip_optimal_control.cpp:147-190), each restated from its own body; theta
 * sections included when o->p > 0 (tm: the theta arena; x-space vectors are
 * then [stagewise x | theta]).  kkt_oracle_add_Kx_to_y above is their
 * composition plus the regularization diagonal (helpers.cpp:953-976), which
 * tests/test_kkt_oracle_reference.py checks.
 * ------------------------------------------------------------------------ */

/* CallbackProvider::add_Hx_to_y, helpers.cpp:978-1067: x, y in x-space. */
void kkt_oracle_add_Hx_to_y(const kkt_oracle *o, const double *model, const double *tm,
                            const double *x, double *y) {
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim;
  for (int i = 0; i < N; ++i) /* :983-992 */
    gemv_n(y + o->voff[KKT_X_STATE][i], BLK(o, model, KKT_NODE_D2L_DX2, i), o->sd[i], o->sd[i],
           x + o->voff[KKT_X_STATE][i], 1.0);
  for (int e = 0; e < E; ++e) { /* :994-1017 */
    const int parent = o->ws.edge_parents[e], n = o->sd[parent], m = o->cd[e];
    const double *xp = x + o->voff[KKT_X_STATE][parent], *ue = x + o->voff[KKT_X_CONTROL][e];
    double *yp = y + o->voff[KKT_X_STATE][parent], *yu = y + o->voff[KKT_X_CONTROL][e];
    gemv_n(yp, BLK(o, model, KKT_EDGE_D2L_DX2, e), n, n, xp, 1.0);
    gemv_n(yp, BLK(o, model, KKT_EDGE_D2L_DXDU, e), n, m, ue, 1.0);
    gemv_t(yu, BLK(o, model, KKT_EDGE_D2L_DXDU, e), n, m, xp);
    gemv_n(yu, BLK(o, model, KKT_EDGE_D2L_DU2, e), m, m, ue, 1.0);
  }
  if (p <= 0) /* :1019-1021 */
    return;
  const double *theta = x + sx;
  double *y_theta = y + sx;
  for (int i = 0; i < N; ++i) { /* :1029-1041 */
    const int n = o->sd[i];
    const double *Hxt = TBLK(o, tm, KKT_TH_NODE_DXDTH, i);
    gemv_n(y + o->voff[KKT_X_STATE][i], Hxt, n, p, theta, 1.0);
    gemv_t(y_theta, Hxt, n, p, x + o->voff[KKT_X_STATE][i]);
    gemv_n(y_theta, TBLK(o, tm, KKT_TH_NODE_DTH2, i), p, p, theta, 1.0);
  }
  for (int e = 0; e < E; ++e) { /* :1042-1066 */
    const int parent = o->ws.edge_parents[e], n = o->sd[parent], m = o->cd[e];
    const double *Hxt = TBLK(o, tm, KKT_TH_EDGE_DXDTH, e), *Hut = TBLK(o, tm, KKT_TH_EDGE_DUDTH, e);
    gemv_n(y + o->voff[KKT_X_STATE][parent], Hxt, n, p, theta, 1.0);
    gemv_n(y + o->voff[KKT_X_CONTROL][e], Hut, m, p, theta, 1.0);
    gemv_t(y_theta, Hxt, n, p, x + o->voff[KKT_X_STATE][parent]);
    gemv_t(y_theta, Hut, m, p, x + o->voff[KKT_X_CONTROL][e]);
    gemv_n(y_theta, TBLK(o, tm, KKT_TH_EDGE_DTH2, e), p, p, theta, 1.0);
  }
}

/* CallbackProvider::add_Cx_to_y, helpers.cpp:1069-1159: x in x-space, y in y-space. */
void kkt_oracle_add_Cx_to_y(const kkt_oracle *o, const double *model, const double *tm,
                            const double *x, double *y) {
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim;
  const int root = o->ws.preorder_nodes[0];
  for (int d = 0; d < o->sd[root]; ++d) /* :1074-1081 */
    y[o->voff[KKT_Y_DYN][root] + d] -= x[o->voff[KKT_X_STATE][root] + d];
  for (int i = 0; i < N; ++i) /* :1083-1093 */
    gemv_n(y + o->voff[KKT_Y_NODE_C][i], BLK(o, model, KKT_NODE_DC_DX, i), o->ncd[i], o->sd[i],
           x + o->voff[KKT_X_STATE][i], 1.0);
  for (int e = 0; e < E; ++e) { /* :1095-1126 */
    const int parent = o->ws.edge_parents[e], child = o->ws.edge_children[e];
    const int np = o->sd[parent], nc = o->sd[child], m = o->cd[e], c = o->ecd[e];
    const double *xp = x + o->voff[KKT_X_STATE][parent], *ue = x + o->voff[KKT_X_CONTROL][e];
    double *yd = y + o->voff[KKT_Y_DYN][child], *yc = y + o->voff[KKT_Y_EDGE_C][e];
    gemv_n(yd, BLK(o, model, KKT_EDGE_DDYN_DX, e), nc, np, xp, 1.0);
    gemv_n(yd, BLK(o, model, KKT_EDGE_DDYN_DU, e), nc, m, ue, 1.0);
    for (int d = 0; d < nc; ++d)
      yd[d] -= x[o->voff[KKT_X_STATE][child] + d];
    gemv_n(yc, BLK(o, model, KKT_EDGE_DC_DX, e), c, np, xp, 1.0);
    gemv_n(yc, BLK(o, model, KKT_EDGE_DC_DU, e), c, m, ue, 1.0);
  }
  if (p <= 0)
    return;
  const double *theta = x + sx;
  for (int i = 0; i < N; ++i) /* :1135-1143 */
    gemv_n(y + o->voff[KKT_Y_NODE_C][i], TBLK(o, tm, KKT_TH_NODE_DC, i), o->ncd[i], p, theta, 1.0);
  for (int e = 0; e < E; ++e) { /* :1144-1158 */
    const int child = o->ws.edge_children[e];
    gemv_n(y + o->voff[KKT_Y_DYN][child], TBLK(o, tm, KKT_TH_EDGE_DDYN, e), o->sd[child], p, theta, 1.0);
    gemv_n(y + o->voff[KKT_Y_EDGE_C][e], TBLK(o, tm, KKT_TH_EDGE_DC, e), o->ecd[e], p, theta, 1.0);
  }
}

/* CallbackProvider::add_CTx_to_y, helpers.cpp:1161-1250: x in y-space, y in x-space. */
void kkt_oracle_add_CTx_to_y(const kkt_oracle *o, const double *model, const double *tm,
                             const double *x, double *y) {
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim;
  const int root = o->ws.preorder_nodes[0];
  for (int d = 0; d < o->sd[root]; ++d) /* :1166-1173 */
    y[o->voff[KKT_X_STATE][root] + d] -= x[o->voff[KKT_Y_DYN][root] + d];
  for (int i = 0; i < N; ++i) /* :1175-1185 */
    gemv_t(y + o->voff[KKT_X_STATE][i], BLK(o, model, KKT_NODE_DC_DX, i), o->ncd[i], o->sd[i],
           x + o->voff[KKT_Y_NODE_C][i]);
  for (int e = 0; e < E; ++e) { /* :1187-1219 */
    const int parent = o->ws.edge_parents[e], child = o->ws.edge_children[e];
    const int np = o->sd[parent], nc = o->sd[child], m = o->cd[e], c = o->ecd[e];
    const double *dyn = x + o->voff[KKT_Y_DYN][child], *cv = x + o->voff[KKT_Y_EDGE_C][e];
    double *yp = y + o->voff[KKT_X_STATE][parent], *ych = y + o->voff[KKT_X_STATE][child];
    double *yu = y + o->voff[KKT_X_CONTROL][e];
    gemv_t(yp, BLK(o, model, KKT_EDGE_DDYN_DX, e), nc, np, dyn);
    gemv_t(yp, BLK(o, model, KKT_EDGE_DC_DX, e), c, np, cv);
    for (int d = 0; d < nc; ++d)
      ych[d] -= dyn[d];
    gemv_t(yu, BLK(o, model, KKT_EDGE_DDYN_DU, e), nc, m, dyn);
    gemv_t(yu, BLK(o, model, KKT_EDGE_DC_DU, e), c, m, cv);
  }
  if (p <= 0)
    return;
  double *y_theta = y + sx;
  for (int i = 0; i < N; ++i) /* :1228-1236 */
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_NODE_DC, i), o->ncd[i], p, x + o->voff[KKT_Y_NODE_C][i]);
  for (int e = 0; e < E; ++e) { /* :1237-1249 */
    const int child = o->ws.edge_children[e];
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_EDGE_DDYN, e), o->sd[child], p, x + o->voff[KKT_Y_DYN][child]);
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_EDGE_DC, e), o->ecd[e], p, x + o->voff[KKT_Y_EDGE_C][e]);
  }
}

/* CallbackProvider::add_Gx_to_y, helpers.cpp:1252-1309: x in x-space, y in z-space. */
void kkt_oracle_add_Gx_to_y(const kkt_oracle *o, const double *model, const double *tm,
                            const double *x, double *y) {
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim;
  for (int i = 0; i < N; ++i) /* :1257-1267 */
    gemv_n(y + o->voff[KKT_Z_NODE][i], BLK(o, model, KKT_NODE_DG_DX, i), o->ngd[i], o->sd[i],
           x + o->voff[KKT_X_STATE][i], 1.0);
  for (int e = 0; e < E; ++e) { /* :1268-1283 */
    const int parent = o->ws.edge_parents[e], n = o->sd[parent], m = o->cd[e], g = o->egd[e];
    gemv_n(y + o->voff[KKT_Z_EDGE][e], BLK(o, model, KKT_EDGE_DG_DX, e), g, n, x + o->voff[KKT_X_STATE][parent], 1.0);
    gemv_n(y + o->voff[KKT_Z_EDGE][e], BLK(o, model, KKT_EDGE_DG_DU, e), g, m, x + o->voff[KKT_X_CONTROL][e], 1.0);
  }
  if (p <= 0)
    return;
  const double *theta = x + sx;
  for (int i = 0; i < N; ++i) /* :1292-1299 */
    gemv_n(y + o->voff[KKT_Z_NODE][i], TBLK(o, tm, KKT_TH_NODE_DG, i), o->ngd[i], p, theta, 1.0);
  for (int e = 0; e < E; ++e) /* :1300-1308 */
    gemv_n(y + o->voff[KKT_Z_EDGE][e], TBLK(o, tm, KKT_TH_EDGE_DG, e), o->egd[e], p, theta, 1.0);
}

/* CallbackProvider::add_GTx_to_y, helpers.cpp:1311-1368: x in z-space, y in x-space. */
void kkt_oracle_add_GTx_to_y(const kkt_oracle *o, const double *model, const double *tm,
                             const double *x, double *y) {
  const int E = o->E, N = o->N, p = o->p, sx = o->x_dim;
  for (int i = 0; i < N; ++i) /* :1316-1326 */
    gemv_t(y + o->voff[KKT_X_STATE][i], BLK(o, model, KKT_NODE_DG_DX, i), o->ngd[i], o->sd[i],
           x + o->voff[KKT_Z_NODE][i]);
  for (int e = 0; e < E; ++e) { /* :1327-1342 */
    const int parent = o->ws.edge_parents[e], n = o->sd[parent], m = o->cd[e], g = o->egd[e];
    gemv_t(y + o->voff[KKT_X_STATE][parent], BLK(o, model, KKT_EDGE_DG_DX, e), g, n, x + o->voff[KKT_Z_EDGE][e]);
    gemv_t(y + o->voff[KKT_X_CONTROL][e], BLK(o, model, KKT_EDGE_DG_DU, e), g, m, x + o->voff[KKT_Z_EDGE][e]);
  }
  if (p <= 0)
    return;
  double *y_theta = y + sx;
  for (int i = 0; i < N; ++i) /* :1351-1358 */
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_NODE_DG, i), o->ngd[i], p, x + o->voff[KKT_Z_NODE][i]);
  for (int e = 0; e < E; ++e) /* :1359-1367 */
    gemv_t(y_theta, TBLK(o, tm, KKT_TH_EDGE_DG, e), o->egd[e], p, x + o->voff[KKT_Z_EDGE][e]);
}
