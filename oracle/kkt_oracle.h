/*
 * kkt_oracle.h -- CPU restatement of the reference's Newton-KKT callbacks
 * (the caller either side of the Riccati path).  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, in plain C99 on top of lqr_oracle.c:
 *   CallbackProvider::factor            helpers.cpp:242-370
 *   CallbackProvider::solve             helpers.cpp:749-893 (single rhs)
 *   CallbackProvider::add_Kx_to_y       helpers.cpp:953-976
 *     add_Hx/Cx/CTx/Gx/GTx_to_y         helpers.cpp:978-1368
 * and the flattened variable ordering of populate_workspace_metadata
 * (types.cpp:24-64); the theta (global variable) Schur complement
 * (helpers.cpp:190-240, 372-407, 896-951) in the *_theta functions below.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Parity pinning: the reference's own tests of this path
 * (tests/variable_dimensions_test.cpp:135-181, 265-336) are re-expressed in
 * tests/test_kkt_oracle_reference.py (same model data, same regularization,
 * K * solution == rhs to 1e-9; :338-363 with theta_dim = 2 to 1e-8), and the
 * solution is compared with an
 * independent numpy assembly + dense solve of the full KKT matrix
 * (oracle/dense_kkt.py: full_kkt_matrix).
 *
 * Model arena of ONE problem (doubles, column-major compact blocks, the
 * fields of NodeModelCallbackOutput / EdgeModelCallbackOutput that the path
 * reads, types.hpp:40-91), node i followed by edge i:
 *   node i : d2L_dx2 (n_i x n_i) | dc_dx (c_i x n_i) | dg_dx (g_i x n_i)
 *   edge e : d2L_dx2 (np x np) | d2L_dxdu (np x m) | d2L_du2 (m x m) |
 *            ddyn_dx (nc x np) | ddyn_du (nc x m) |
 *            dc_dx (ce x np) | dc_du (ce x m) | dg_dx (ge x np) | dg_du (ge x m)
 *   np / nc = state dimension of the edge's parent / child node.
 * KKT vectors are [x | y | z] with the reference's offsets:
 *   x: per node i: state_i, then (i < E) control_i
 *   y: per node i: dyn_i (n_i), node_c_i; then per edge: edge_c_e
 *   z: per node i: node_g_i; then per edge: edge_g_e
 */
#ifndef KKT_ORACLE_H
#define KKT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* status of factor: 0 == the reference's `true`; anything else == `false`.
 * 1..4 are LQR::FactorStatus (lqr.hpp:68-74). */
enum {
  KKT_ORACLE_SUCCESS = 0,
  KKT_ORACLE_NONPOSITIVE_REGULARIZATION = 5, /* helpers.cpp:256-300 */
  KKT_ORACLE_INVALID_INPUT = 6               /* helpers.cpp:244-246 */
};

/* blocks for kkt_oracle_model_offset() */
enum {
  KKT_NODE_D2L_DX2 = 0, KKT_NODE_DC_DX, KKT_NODE_DG_DX,
  KKT_EDGE_D2L_DX2, KKT_EDGE_D2L_DXDU, KKT_EDGE_D2L_DU2, KKT_EDGE_DDYN_DX, KKT_EDGE_DDYN_DU,
  KKT_EDGE_DC_DX, KKT_EDGE_DC_DU, KKT_EDGE_DG_DX, KKT_EDGE_DG_DU,
  KKT_NUM_BLOCKS
};
/* tables for kkt_oracle_vector_offset(): types.cpp:33-63 */
enum {
  KKT_X_STATE = 0, KKT_X_CONTROL, KKT_Y_DYN, KKT_Y_NODE_C, KKT_Y_EDGE_C, KKT_Z_NODE, KKT_Z_EDGE
};

typedef struct kkt_oracle kkt_oracle;

/* Any of the four constraint-dimension arrays may be NULL (all zero), like
 * Dimensions::get_node_c_dim & co (lqr.cpp:98-112). */
kkt_oracle *kkt_oracle_create(int num_edges, int root, const int *edge_parents,
                              const int *edge_children, const int *state_dims,
                              const int *control_dims, const int *node_c_dims,
                              const int *node_g_dims, const int *edge_c_dims,
                              const int *edge_g_dims);
void kkt_oracle_destroy(kkt_oracle *o);

/* which: 0 x_dim, 1 y_dim, 2 z_dim, 3 model arena length */
long kkt_oracle_dim(const kkt_oracle *o, int which);
long kkt_oracle_model_offset(const kkt_oracle *o, int block, int index);
long kkt_oracle_vector_offset(const kkt_oracle *o, int table, int index);

int kkt_oracle_factor(kkt_oracle *o, const double *model, const double *w,
                      const double *r1, const double *r2, const double *r3);
void kkt_oracle_solve(kkt_oracle *o, const double *model, const double *b,
                      double *sol);
/* y += K x with K = [[H + r1, C^T, G^T], [C, -r2, 0], [G, 0, -(w + r3)]];
 * x and y are [x | y | z] vectors. */
void kkt_oracle_add_Kx_to_y(const kkt_oracle *o, const double *model,
                            const double *w, const double *r1, const double *r2,
                            const double *r3, const double *x, double *y);

/* The LQR problem the last factor() condensed to (RegularizedLQRData,
 * types.hpp:142-153): block `name` in "QMRqrcd" (d = dyn_r2) of node/edge
 * `index`; for tests of the condensation alone. */
const double *kkt_oracle_lqr_block(const kkt_oracle *o, char name, int index);

/*
 * Global variables theta (Dimensions::theta_dim = p > 0): the Schur-complement
 * path of CallbackProvider::factor (helpers.cpp:372-407, with
 * form_theta_jacobian :190-240 and the multi-right-hand-side
 * solve_stagewise_kkt_matrix :414-747), CallbackProvider::solve (:896-951) and
 * the theta terms of add_*x_to_y (:1023-1066, 1128-1158, 1221-1249, 1285-1308,
 * 1344-1367).  The multi-rhs block of the reference computes, column by column
 * of J_theta, exactly the quantities of its single-rhs block (GEMM in place of
 * GEMV), so it is restated as a loop over columns.
 *
 * Theta arena of ONE problem (doubles, column-major blocks, node i then edge i):
 *   node i : d2L_dxdtheta (n x p) | dc_dtheta (c x p) | dg_dtheta (g x p) |
 *            d2L_dtheta2 (p x p)
 *   edge e : d2L_dxdtheta (np x p) | d2L_dudtheta (m x p) | ddyn_dtheta (nc x p) |
 *            dc_dtheta (ce x p) | dg_dtheta (ge x p) | d2L_dtheta2 (p x p)
 * Vectors with theta: x = [stagewise x | theta], i.e. b, sol = [x | theta | y | z];
 * r1 has x_dim + p entries (theta last).
 */
enum {
  KKT_TH_NODE_DXDTH = 0, KKT_TH_NODE_DC, KKT_TH_NODE_DG, KKT_TH_NODE_DTH2,
  KKT_TH_EDGE_DXDTH, KKT_TH_EDGE_DUDTH, KKT_TH_EDGE_DDYN, KKT_TH_EDGE_DC, KKT_TH_EDGE_DG, KKT_TH_EDGE_DTH2,
  KKT_TH_NUM_BLOCKS
};
#define KKT_ORACLE_THETA_SCHUR_FAILURE 7 /* helpers.cpp:404-407: LLT of the Schur complement failed */

/* Enables theta_dim = p on an oracle (call once, before factor). */
void kkt_oracle_set_theta(kkt_oracle *o, int theta_dim);
/* which: 0 theta arena length */
long kkt_oracle_theta_len(const kkt_oracle *o);
long kkt_oracle_theta_offset(const kkt_oracle *o, int block, int index);
int kkt_oracle_factor_theta(kkt_oracle *o, const double *model, const double *theta_model,
                            const double *w, const double *r1, const double *r2, const double *r3);
void kkt_oracle_solve_theta(kkt_oracle *o, const double *model, const double *theta_model,
                            const double *b, double *sol);
void kkt_oracle_add_Kx_to_y_theta(const kkt_oracle *o, const double *model, const double *theta_model,
                                  const double *w, const double *r1, const double *r2, const double *r3,
                                  const double *x, double *y);

/* The five block operators of CallbackProvider (helpers.hpp:20-24, bodies helpers.cpp:978-1368),
 * y += (block) x, each restated from its own body.  Vector spaces: x-space = [stagewise x | theta
 * (p entries, when kkt_oracle_set_theta was called)], y-space = y_dim, z-space = z_dim entries.
 *   Hx : x-space -> x-space    Cx : x-space -> y-space    CTx : y-space -> x-space
 *   Gx : x-space -> z-space    GTx: z-space -> x-space
 * theta_model: the theta arena, ignored (may be NULL) while theta_dim == 0. */
void kkt_oracle_add_Hx_to_y(const kkt_oracle *o, const double *model, const double *theta_model,
                            const double *x, double *y);
void kkt_oracle_add_Cx_to_y(const kkt_oracle *o, const double *model, const double *theta_model,
                            const double *x, double *y);
void kkt_oracle_add_CTx_to_y(const kkt_oracle *o, const double *model, const double *theta_model,
                             const double *x, double *y);
void kkt_oracle_add_Gx_to_y(const kkt_oracle *o, const double *model, const double *theta_model,
                            const double *x, double *y);
void kkt_oracle_add_GTx_to_y(const kkt_oracle *o, const double *model, const double *theta_model,
                             const double *x, double *y);

/* factor + solve of `batch` problems sharing the topology (arenas strided by
 * the per-problem lengths), OpenMP over problems with one handle per thread.
 * status[p] as kkt_oracle_factor. */
int kkt_oracle_batch(const kkt_oracle *o, long batch, const double *model,
                     const double *w, const double *r1, const double *r2,
                     const double *r3, const double *b, double *sol,
                     int *status, int threads);

#ifdef __cplusplus
}
#endif
#endif
