/*
 * lqr_oracle.h -- CPU restatement of the reference's regularized tree-LQR
 * (Riccati) solver.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X HIP path.  It restates, in plain
 * C99 with hand-written dense kernels (no Eigen), the algorithm of
 *   /root/reference/sip_optimal_control/lqr.cpp:473-871
 * (compute_delta_sqrt, factor_F, compute_regularized_W, F_inv_mult_vector,
 * compile_topology_data, LQR::factor_with_status, LQR::solve) behind the data
 * model of /root/reference/sip_optimal_control/lqr.hpp:5-200.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (sip_optimal_control_amd/) never does.
 *
 * Parity pinning: the reference cannot be compiled in the build container
 * (Eigen 3.4.0, googletest, google-benchmark, Bazel are un-vendored network
 * dependencies, MODULE.bazel:8-24), and it ships no golden vectors.  The
 * oracle is therefore pinned by re-expressing every known-answer / property
 * test of /root/reference/tests/lqr_test.cpp against it
 * (tests/test_oracle_reference_kats.py) and by an independent numpy dense-KKT
 * solve (oracle/dense_kkt.py, the construction of lqr_test.cpp:859-929).
 *
 * All matrices are column-major, compact (leading dimension = rows), exactly
 * as the reference maps them (Eigen::Map<MatrixXd>(ptr, rows, cols)).
 */
#ifndef LQR_ORACLE_H
#define LQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* lqr.hpp:68-74 */
enum {
  LQR_ORACLE_SUCCESS = 0,
  LQR_ORACLE_INVALID_DELTA = 1,
  LQR_ORACLE_F_FACTORIZATION_FAILURE = 2,
  LQR_ORACLE_G_FACTORIZATION_FAILURE = 3,
  LQR_ORACLE_INVALID_TOPOLOGY = 4
};

/* Problem view: lqr.hpp:5-22 (Topology), :24-64 (Dimensions), :76-89 (Input).
 * Non-owning.  Q,q,c,delta are indexed by node; M,R,r,A,B by edge. */
typedef struct {
  int num_edges;
  int root;
  const int *edge_parents;
  const int *edge_children;
  const int *state_dims;   /* [num_edges+1] */
  const int *control_dims; /* [num_edges]   */
  double **Q, **M, **R, **q, **r, **A, **B, **c, **delta;
} lqr_oracle_problem;

/* Factor state: lqr.hpp:109-135.  Same fields, same meaning. */
typedef struct {
  double **W, **K, **V, **G_factor, **F_factor;
  double **sqrt_delta, **sqrt_delta_inv, **k, **v;
  double *G, *g, *H, *h, *F, *f;
  int *child_offsets, *child_edges, *edge_parents, *edge_children;
  int *preorder_nodes, *postorder_nodes, *node_marks;
  int traversal_status;
  int num_edges;
} lqr_oracle_workspace;

/* lqr.cpp:223-272 (reserve) / :274-319 (free). Returns 0 on success. */
int lqr_oracle_workspace_reserve(lqr_oracle_workspace *ws,
                                 const lqr_oracle_problem *p);
void lqr_oracle_workspace_free(lqr_oracle_workspace *ws);

/* lqr.cpp:563-631; result latched in ws->traversal_status (lqr.cpp:640-643).
 */
int lqr_oracle_compile_topology(const lqr_oracle_problem *p,
                                lqr_oracle_workspace *ws);
/* lqr.cpp:645-731 */
int lqr_oracle_factor(const lqr_oracle_problem *p, lqr_oracle_workspace *ws);
/* lqr.cpp:735-871 */
void lqr_oracle_solve(const lqr_oracle_problem *p, lqr_oracle_workspace *ws,
                      double **x, double **u, double **y);

/*
 * Batched uniform-chain convenience used by the parity tests and by the
 * cpu_baseline leg of bench.py.  It runs the functions above, one problem
 * after the other (threads > 1: an OpenMP loop over problems, one workspace
 * per thread), on the packed layout that the HIP library uses on the device
 * (see include/sip_lqr_amd.h, "Packed chain layout"):
 *
 *   mats[p] : for i in 0..T : Q_i (n*n) | delta_i (n)
 *                             then, if i < T: A_i (n*n) | B_i (n*m) |
 *                             M_i (n*m) | R_i (m*m)
 *   vecs[p] : for i in 0..T : q_i (n) | c_i (n) then, if i < T: r_i (m)
 *   sol[p]  : for i in 0..T : x_i (n) | y_i (n) then, if i < T: u_i (m)
 *   gains[p]: for i in 0..T-1 : K_i (m*n, col-major) | k_i (m)
 *   status[p]: FactorStatus of problem p; sol/gains of a failed problem are
 *              left untouched.
 * Scalars are double.  Returns 0.
 */
int lqr_oracle_chain_batch(int n, int m, int T, long batch, const double *mats,
                           const double *vecs, double *sol, double *gains,
                           int *status, int threads);

/* Sizes (in scalars) of one problem in the packed chain layout. */
long lqr_oracle_chain_mats_len(int n, int m, int T);
long lqr_oracle_chain_vecs_len(int n, int m, int T);
long lqr_oracle_chain_gains_len(int n, int m, int T);

#ifdef __cplusplus
}
#endif
#endif
