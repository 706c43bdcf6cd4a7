/*
 * sip_kkt_amd.h -- C ABI of the batched Newton-KKT step on MI355X (gfx950):
 * the caller either side of the Riccati path (SURVEY.md section 8, rows f1
 * and f4), device-resident, for `batch` problems that share one topology and
 * dimension table.
 *
 * Replaces, of /root/reference/sip_optimal_control (class CallbackProvider,
 * helpers.hpp:7-33):
 *   factor(w, r1, r2, r3)   helpers.cpp:242-370   -> sip_kkt_factor
 *   solve(b, sol)           helpers.cpp:749-893   -> sip_kkt_solve
 *   factor + solve          newton_kkt_benchmark.cpp:316-324 (loop body of
 *                           BM_NewtonKKTFactorSolve)  -> sip_kkt_factor_solve
 *   add_Kx_to_y(...)        helpers.cpp:953-976   -> sip_kkt_add_Kx_to_y
 *   add_Hx_to_y(x, y)       helpers.cpp:978-1067  -> sip_kkt_add_Hx_to_y
 *   add_Cx_to_y(x, y)       helpers.cpp:1069-1159 -> sip_kkt_add_Cx_to_y
 *   add_CTx_to_y(x, y)      helpers.cpp:1161-1250 -> sip_kkt_add_CTx_to_y
 *   add_Gx_to_y(x, y)       helpers.cpp:1252-1309 -> sip_kkt_add_Gx_to_y
 *   add_GTx_to_y(x, y)      helpers.cpp:1311-1368 -> sip_kkt_add_GTx_to_y
 *                           (the callbacks SIP is handed one by one,
 *                           sip_optimal_control.cpp:147-190)
 * The Riccati solve in the middle is the library's own LQR path
 * (sip_lqr_amd.h): uniform chains use the packed chain layout and its fused
 * kernels, everything else the general tree engine.
 *
 * Global variables theta (theta_dim > 0: the Schur complement and its
 * multi-right-hand-side solve, helpers.cpp:372-747, 896-951): the *_theta
 * entry points at the end of this header.
 *
 * Data (all double, device memory, problem p at p * per-problem length):
 *
 *  model  [sip_kkt_len(plan, SIP_KKT_LEN_MODEL)] per problem -- the fields of
 *    NodeModelCallbackOutput / EdgeModelCallbackOutput (types.hpp:48-89)
 *    that the path reads, column-major compact blocks, node i followed by
 *    edge i:
 *      node i : d2L_dx2 (n_i x n_i) | dc_dx (c_i x n_i) | dg_dx (g_i x n_i)
 *      edge e : d2L_dx2 (np x np) | d2L_dxdu (np x m) | d2L_du2 (m x m) |
 *               ddyn_dx (nc x np) | ddyn_du (nc x m) | dc_dx (ce x np) |
 *               dc_du (ce x m) | dg_dx (ge x np) | dg_du (ge x m)
 *    (np / nc: state dimension of the edge's parent / child), block offsets
 *    from sip_kkt_model_offset().
 *  KKT vectors: the reference's flattened ordering (types.cpp:24-64,
 *    offsets from sip_kkt_vector_offset()):
 *      x [x_dim]: per node i: state_i, then (i < num_edges) control_i
 *      y [y_dim]: per node i: dyn_i (n_i) | node_c_i; then per edge edge_c_e
 *      z [z_dim]: per node i: node_g_i; then per edge edge_g_e
 *    r1 [x_dim], r2 [y_dim], w and r3 [z_dim]; b and sol [x_dim+y_dim+z_dim]
 *    = [x | y | z].
 *
 * Error conventions: functions return SIP_LQR_OK or a SIP_LQR_ERR_* code of
 * sip_lqr_amd.h for API misuse / HIP failures.  Numerical outcomes are
 * per-problem int32 status words: 0 where the reference's factor() returns
 * true; otherwise the reason it returned false (values below).  sol of a
 * problem whose status != 0 is left untouched.  No host fallback.
 */
#ifndef SIP_KKT_AMD_H
#define SIP_KKT_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "sip_lqr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* status[p]: 0..4 = LQR::FactorStatus (lqr.hpp:68-74) from the Riccati
 * factorization; the rest are the other `return false` exits of
 * CallbackProvider::factor. */
enum {
  SIP_KKT_SUCCESS = 0,
  SIP_KKT_NONPOSITIVE_REGULARIZATION = 5, /* r2 <= 0 or w + r3 <= 0, helpers.cpp:256-297 */
  SIP_KKT_INVALID_INPUT = 6               /* validate_input() failed, helpers.cpp:244-246 */
};

/* per-problem lengths in doubles for sip_kkt_len() */
enum {
  SIP_KKT_LEN_X = 0,  /* Dimensions::get_x_dim, lqr.cpp:153-155 */
  SIP_KKT_LEN_Y = 1,  /* Dimensions::get_y_dim, lqr.cpp:157-165 */
  SIP_KKT_LEN_Z = 2,  /* Dimensions::get_z_dim, lqr.cpp:167-175 */
  SIP_KKT_LEN_MODEL = 3
};

/* blocks of the model arena for sip_kkt_model_offset() */
enum {
  SIP_KKT_NODE_D2L_DX2 = 0, SIP_KKT_NODE_DC_DX, SIP_KKT_NODE_DG_DX,
  SIP_KKT_EDGE_D2L_DX2, SIP_KKT_EDGE_D2L_DXDU, SIP_KKT_EDGE_D2L_DU2,
  SIP_KKT_EDGE_DDYN_DX, SIP_KKT_EDGE_DDYN_DU, SIP_KKT_EDGE_DC_DX,
  SIP_KKT_EDGE_DC_DU, SIP_KKT_EDGE_DG_DX, SIP_KKT_EDGE_DG_DU,
  SIP_KKT_NUM_BLOCKS
};

/* offset tables of the flattened KKT vectors (Workspace::x_state_offsets
 * ... z_edge_offsets, types.cpp:33-63) for sip_kkt_vector_offset() */
enum {
  SIP_KKT_X_STATE = 0, SIP_KKT_X_CONTROL, SIP_KKT_Y_DYN, SIP_KKT_Y_NODE_C,
  SIP_KKT_Y_EDGE_C, SIP_KKT_Z_NODE, SIP_KKT_Z_EDGE
};

typedef struct sip_kkt_plan sip_kkt_plan;

/* Replaces: CallbackProvider::CallbackProvider (helpers.cpp:11-26) +
 * validate_input (types.cpp:68-127) + populate_workspace_metadata
 * (types.cpp:24-64).  Any of the four constraint-dimension arrays may be NULL
 * (all zero), as Dimensions::get_node_c_dim & co allow (lqr.cpp:98-112).  An
 * invalid input is latched: the plan is created and every factor call reports
 * SIP_KKT_INVALID_INPUT (or the traversal's INVALID_TOPOLOGY) for all
 * problems, like input_is_valid_ (helpers.cpp:24-26, 244-246). */
int sip_kkt_plan_create(int64_t batch, int num_edges, int root,
                        const int *edge_parents, const int *edge_children,
                        const int *state_dims, const int *control_dims,
                        const int *node_c_dims, const int *node_g_dims,
                        const int *edge_c_dims, const int *edge_g_dims,
                        int device, sip_kkt_plan **plan);
void sip_kkt_plan_destroy(sip_kkt_plan *plan);

/* 0 if the input validated, else the status every factor call will report. */
int sip_kkt_input_status(const sip_kkt_plan *plan);
size_t sip_kkt_len(const sip_kkt_plan *plan, int which);
/* (size_t)-1 for an out-of-range request */
size_t sip_kkt_model_offset(const sip_kkt_plan *plan, int block, int index);
size_t sip_kkt_vector_offset(const sip_kkt_plan *plan, int table, int index);
/* Bytes of device scratch for the whole batch: the condensed LQR problem
 * (Workspace::RegularizedLQRData, types.hpp:163-176: Q_mod, M_mod, R_mod,
 * q_mod, r_mod, c_mod, dyn_r2, the 1/r2 and 1/(w+r3) weights), the LQR
 * output and the Riccati factor state (LQR::Workspace, lqr.hpp:109-135). */
size_t sip_kkt_work_bytes(const sip_kkt_plan *plan);
/* which Riccati path and which kernel variants the plan runs, e.g.
 * "chain:chain_factor_solve_qw16<...> + chain condensation (A|B in place, Q|R packed) [benchmark-family instantiation]"
 * or "tree:general".  The string belongs to the plan; sip_kkt_plan_set_theta appends to it (" + fused theta passes"),
 * so a pointer obtained before that call must be fetched again after it. */
const char *sip_kkt_kernel_name(const sip_kkt_plan *plan);

/* Replaces CallbackProvider::factor (helpers.cpp:242-370): checks and inverts
 * the regularization, condenses the constraint Jacobians into Q_mod / M_mod /
 * R_mod (rank-c / rank-g symmetric updates, helpers.cpp:79-136, 299-361) and
 * runs the Riccati factorization.  Writes status[p].  Asynchronous on
 * `stream`. */
int sip_kkt_factor(const sip_kkt_plan *plan, const double *d_model,
                   const double *d_w, const double *d_r1, const double *d_r2,
                   const double *d_r3, void *d_work, int32_t *d_status,
                   void *stream);

/* Replaces CallbackProvider::solve (helpers.cpp:896-900 -> 749-893) against
 * the last sip_kkt_factor() on the same d_work: builds q_mod / r_mod / c_mod
 * from b, runs LQR::solve, recovers the constraint multipliers y_c, z.  May
 * be called repeatedly with new right-hand sides.  d_status: as written by
 * sip_kkt_factor (problems with status != 0 are skipped). */
int sip_kkt_solve(const sip_kkt_plan *plan, const double *d_model,
                  const double *d_b, double *d_sol, void *d_work,
                  const int32_t *d_status, void *stream);

/* factor() followed by solve() (the loop body of BM_NewtonKKTFactorSolve,
 * benchmarks/newton_kkt_benchmark.cpp:316-324), with the Riccati part as the
 * fused launch of sip_lqr_factor_solve() when the plan is a uniform chain
 * (sip_lqr_factor_solve_split where the plan's kernel has that form: the
 * dynamics Jacobians are then read in place, not copied, helpers.cpp:365-366).
 * It does NOT leave the factor state sip_kkt_solve() works from -- the fused
 * sweep keeps none, and d_work holds the inputs in the fused sweep's own
 * layout: for further right-hand sides on one factorization call
 * sip_kkt_factor() and then sip_kkt_solve() as often as needed. */
int sip_kkt_factor_solve(const sip_kkt_plan *plan, const double *d_model,
                         const double *d_w, const double *d_r1,
                         const double *d_r2, const double *d_r3,
                         const double *d_b, double *d_sol, void *d_work,
                         int32_t *d_status, void *stream);

/* Replaces CallbackProvider::add_Kx_to_y (helpers.cpp:953-976, with
 * add_Hx/Cx/CTx/Gx/GTx_to_y, :978-1368): y += K x for
 *   K = [[H + r1, C^T, G^T], [C, -r2, 0], [G, 0, -(w + r3)]],
 * x and y being [x | y | z] vectors of every problem. */
int sip_kkt_add_Kx_to_y(const sip_kkt_plan *plan, const double *d_model,
                        const double *d_w, const double *d_r1,
                        const double *d_r2, const double *d_r3,
                        const double *d_x, double *d_y, void *stream);

/* Replace CallbackProvider::add_Hx_to_y / add_Cx_to_y / add_CTx_to_y /
 * add_Gx_to_y / add_GTx_to_y (helpers.hpp:20-24; bodies helpers.cpp:978-1368):
 * y += (block) x for one block of K, for every problem of the batch.  Unlike
 * add_Kx_to_y the vectors are per vector space, [batch][len] contiguous:
 *   x-space: the primal variables, len = sip_kkt_len(SIP_KKT_LEN_X)
 *   y-space: the equality multipliers, len = sip_kkt_len(SIP_KKT_LEN_Y)
 *   z-space: the inequality multipliers, len = sip_kkt_len(SIP_KKT_LEN_Z)
 *     Hx : x-space -> x-space      Cx : x-space -> y-space     CTx: y-space -> x-space
 *     Gx : x-space -> z-space      GTx: z-space -> x-space
 * A block with an empty space is a no-op.  add_Kx_to_y is their sum plus the
 * regularization diagonal (helpers.cpp:958-975). */
int sip_kkt_add_Hx_to_y(const sip_kkt_plan *plan, const double *d_model,
                        const double *d_x, double *d_y, void *stream);
int sip_kkt_add_Cx_to_y(const sip_kkt_plan *plan, const double *d_model,
                        const double *d_x, double *d_y, void *stream);
int sip_kkt_add_CTx_to_y(const sip_kkt_plan *plan, const double *d_model,
                         const double *d_x, double *d_y, void *stream);
int sip_kkt_add_Gx_to_y(const sip_kkt_plan *plan, const double *d_model,
                        const double *d_x, double *d_y, void *stream);
int sip_kkt_add_GTx_to_y(const sip_kkt_plan *plan, const double *d_model,
                         const double *d_x, double *d_y, void *stream);

/* ------------------------------------------------------------------------
 * Global variables theta (Dimensions::theta_dim = p > 0; SURVEY.md section 8
 * row f2): the Schur complement on theta around the stagewise solve.
 *
 * Replaces, for p > 0: form_theta_jacobian (helpers.cpp:190-240), the theta
 * part of CallbackProvider::factor (:372-407) with its multi-right-hand-side
 * stagewise solve (:414-747), CallbackProvider::solve (:896-951) and the
 * theta terms of add_*x_to_y (:1023-1066, 1128-1158, 1221-1249, 1285-1308,
 * 1344-1367).  K^-1 J_theta runs as p launches of the plan's solve path, one
 * per column (the reference's multi-rhs block computes the same quantities
 * column by column, GEMM in place of GEMV).
 *
 * theta arena [sip_kkt_theta_len(plan)] per problem (doubles, column-major
 * blocks, node i then edge i):
 *   node i : d2L_dxdtheta (n x p) | dc_dtheta (c x p) | dg_dtheta (g x p) |
 *            d2L_dtheta2 (p x p)
 *   edge e : d2L_dxdtheta (np x p) | d2L_dudtheta (m x p) | ddyn_dtheta (nc x p) |
 *            dc_dtheta (ce x p) | dg_dtheta (ge x p) | d2L_dtheta2 (p x p)
 * Vectors: x = [stagewise x | theta]: r1 has x_dim + p entries, b and sol are
 * [x | theta | y | z] (x_dim + p + y_dim + z_dim).  sip_kkt_len() keeps
 * reporting the stagewise x_dim.
 * ---------------------------------------------------------------------- */
enum {
  SIP_KKT_TH_NODE_D2L_DXDTHETA = 0, SIP_KKT_TH_NODE_DC_DTHETA, SIP_KKT_TH_NODE_DG_DTHETA,
  SIP_KKT_TH_NODE_D2L_DTHETA2, SIP_KKT_TH_EDGE_D2L_DXDTHETA, SIP_KKT_TH_EDGE_D2L_DUDTHETA,
  SIP_KKT_TH_EDGE_DDYN_DTHETA, SIP_KKT_TH_EDGE_DC_DTHETA, SIP_KKT_TH_EDGE_DG_DTHETA,
  SIP_KKT_TH_EDGE_D2L_DTHETA2, SIP_KKT_TH_NUM_BLOCKS
};
/* status[p] when the LLT of the theta Schur complement fails (helpers.cpp:404-407) */
#define SIP_KKT_THETA_SCHUR_FAILURE 7

/* Declares theta_dim = p (> 0) on a valid plan; call once, before the *_theta
 * entry points (Dimensions::theta_dim, lqr.hpp:25). */
int sip_kkt_plan_set_theta(sip_kkt_plan *plan, int theta_dim);
size_t sip_kkt_theta_len(const sip_kkt_plan *plan);
size_t sip_kkt_theta_offset(const sip_kkt_plan *plan, int block, int index);
/* Bytes of the extra device scratch: J_theta, K^-1 J_theta
 * (Workspace::RegularizedLQRData::theta_jacobian / theta_solution,
 * types.hpp:174-175), the Schur factor, theta_rhs and the stagewise rhs /
 * solution (types.hpp:176-179), for the whole batch.  Plans of uniform chains
 * never assemble J_theta (they read its entries from the theta arena where
 * they are needed, csrc/kkt_theta_chain_kernels.hpp) and keep per-stage
 * partial sums of the Schur complement and of J_theta^T K^-1 b in its place:
 * the size reported here says which.  d_theta_work carries K^-1 J_theta and
 * the Schur factor from sip_kkt_factor_theta to sip_kkt_solve_theta. */
size_t sip_kkt_theta_work_bytes(const sip_kkt_plan *plan);

int sip_kkt_factor_theta(const sip_kkt_plan *plan, const double *d_model,
                         const double *d_theta_model, const double *d_w,
                         const double *d_r1, const double *d_r2,
                         const double *d_r3, void *d_work, void *d_theta_work,
                         int32_t *d_status, void *stream);
int sip_kkt_solve_theta(const sip_kkt_plan *plan, const double *d_model,
                        const double *d_theta_model, const double *d_b,
                        double *d_sol, void *d_work, void *d_theta_work,
                        const int32_t *d_status, void *stream);
int sip_kkt_add_Kx_to_y_theta(const sip_kkt_plan *plan, const double *d_model,
                              const double *d_theta_model, const double *d_w,
                              const double *d_r1, const double *d_r2,
                              const double *d_r3, const double *d_x,
                              double *d_y, void *stream);

/* The five block operators with theta: x-space vectors are [stagewise x |
 * theta] (len = sip_kkt_len(SIP_KKT_LEN_X) + theta_dim) and the theta sections
 * of the reference's bodies are included (helpers.cpp:1023-1066, 1128-1158,
 * 1221-1249, 1285-1308, 1344-1367). */
int sip_kkt_add_Hx_to_y_theta(const sip_kkt_plan *plan, const double *d_model,
                              const double *d_theta_model, const double *d_x,
                              double *d_y, void *stream);
int sip_kkt_add_Cx_to_y_theta(const sip_kkt_plan *plan, const double *d_model,
                              const double *d_theta_model, const double *d_x,
                              double *d_y, void *stream);
int sip_kkt_add_CTx_to_y_theta(const sip_kkt_plan *plan, const double *d_model,
                               const double *d_theta_model, const double *d_x,
                               double *d_y, void *stream);
int sip_kkt_add_Gx_to_y_theta(const sip_kkt_plan *plan, const double *d_model,
                              const double *d_theta_model, const double *d_x,
                              double *d_y, void *stream);
int sip_kkt_add_GTx_to_y_theta(const sip_kkt_plan *plan, const double *d_model,
                               const double *d_theta_model, const double *d_x,
                               double *d_y, void *stream);

#ifdef __cplusplus
}
#endif
#endif
