/*
 * sip_lqr_amd.h -- C ABI of the MI355X (gfx950) batched regularized-LQR /
 * Riccati solver.  This is the drop-in boundary for the Newton-KKT linear
 * solve of joaospinto/sip_optimal_control: every entry point names the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * The reference solves ONE problem per `LQR` object on the host
 * (sip_optimal_control/lqr.hpp:66-200).  This library solves `batch`
 * independent problems of one shape (uniform-dimension chain: the topology
 * of Topology::set_chain, lqr.cpp:32-40, and of Dimensions::set_uniform,
 * lqr.cpp:77-88) per call, on the GPU, with device-resident inputs and
 * outputs.  There is no CPU fallback: every compute entry point fails with
 * SIP_LQR_ERR_HIP / SIP_LQR_ERR_UNSUPPORTED rather than compute on the host.
 *
 * Plain C, plain pointers and sizes, no C++/torch types.  All device
 * pointers are ordinary HIP device pointers; `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  The caller owns every byte
 * (as in the reference: lqr.hpp:136-186, "mem_assign"/"num_bytes").
 * Device buffers must be 16-byte aligned (hipMalloc and every framework
 * allocator are; the LDS-DMA kernels return SIP_LQR_ERR_HIP otherwise).
 * The compute entry points only enqueue kernels on `stream` -- no host
 * synchronisation, no allocation, no memset / memcpy nodes -- so a loop of
 * them can be captured in a hipGraph and replayed on any stream, the first
 * call on a plan included: everything a plan needs on the device is uploaded
 * by its *_plan_create (the one exception is a plan latched INVALID at
 * creation, which reports through a synchronous copy).
 * Devices: a plan belongs to the HIP device ordinal it was created with.  Every
 * compute entry point makes that device current for the duration of the call
 * and restores the caller's current device before it returns, so `stream`
 * must be a stream of the plan's device (NULL = that device's default stream)
 * and the buffers must live on it (or be peer-accessible).  *_plan_create
 * leaves the caller's current device untouched as well.  A chain plan
 * created on a host without any HIP device serves the host-side entry points
 * only (sizes, pack / unpack, kernel name); its compute entry points return
 * SIP_LQR_ERR_HIP.
 *
 * ------------------------------------------------------------------------
 * Packed chain layout (device and host-staging buffers; scalar = double for
 * SIP_LQR_F64, float for SIP_LQR_F32).  All blocks are column-major and
 * compact (ld = rows), exactly as the reference maps them
 * (Eigen::Map<MatrixXd>(ptr, rows, cols), e.g. lqr.cpp:654-687).  Problems
 * are stored one after the other (problem-major); inside a problem, stages
 * i = 0..T follow each other; node i is the parent of edge i, node i+1 its
 * child:
 *
 *   mats  [batch][ for i in 0..T :  Q_i (n*n) | delta_i (n)          "node"
 *                    and, if i < T: A_i (n*n) | B_i (n*m) |
 *                                   M_i (n*m) | R_i (m*m) ]          "edge"
 *   vecs  [batch][ for i in 0..T :  q_i (n) | c_i (n)   , if i<T: r_i (m) ]
 *   sol   [batch][ for i in 0..T :  x_i (n) | y_i (n)   , if i<T: u_i (m) ]
 *   gains [batch][ for i in 0..T-1: K_i (m*n) | k_i (m) ]
 *   status[batch]  int32, FactorStatus codes below
 *
 * mats holds what LQR::factor() reads (LQR::Input::{Q,M,R,A,B,delta},
 * lqr.hpp:77-85; patched in by helpers.cpp:362-367), vecs what LQR::solve()
 * additionally reads ({q,r,c}, helpers.cpp:814-816), sol what it writes
 * (LQR::Output::{x,u,y}, lqr.hpp:91-94), gains the feedback terms
 * LQR::Workspace::{K,k} (lqr.hpp:112,118).
 * ------------------------------------------------------------------------
 */
#ifndef SIP_LQR_AMD_H
#define SIP_LQR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-problem factor status: the values of LQR::FactorStatus,
 * sip_optimal_control/lqr.hpp:68-74. */
enum {
  SIP_LQR_SUCCESS = 0,
  SIP_LQR_INVALID_DELTA = 1,
  SIP_LQR_F_FACTORIZATION_FAILURE = 2,
  SIP_LQR_G_FACTORIZATION_FAILURE = 3,
  SIP_LQR_INVALID_TOPOLOGY = 4
};

/* API-level return codes (the reference has no error channel besides
 * FactorStatus; these report misuse of this library). */
enum {
  SIP_LQR_OK = 0,
  SIP_LQR_ERR_INVALID_ARGUMENT = -1,
  SIP_LQR_ERR_UNSUPPORTED = -2, /* shape/dtype has no HIP kernel */
  SIP_LQR_ERR_HIP = -3,         /* a HIP runtime call failed   */
  SIP_LQR_ERR_ALLOC = -4
};

enum { SIP_LQR_F64 = 0, SIP_LQR_F32 = 1 };
/* Layout of `mats` (see "Packed chain layout" above):
 *   FULL       Q (n x n) and R (m x m) as full column-major squares, as LQR::Input holds them;
 *   SYMMETRIC  Q and R as their lower triangles packed by columns (column c: rows c .. n-1): n (n + 1) / 2 and
 *              m (m + 1) / 2 scalars.  The reference reads both triangles of a Q it takes to be symmetric
 *              (`V = Q`, lqr.cpp:658) and only the lower one of R (Eigen::LLT of G, lqr.cpp:697 -- G's strictly
 *              upper part comes from R + (B^T W) B there, the kernels here use the symmetric counterpart), so the
 *              upper triangles are redundant bytes of a bandwidth-bound sweep: 576 of the 7 840 B a problem-stage
 *              moves at n = 12, m = 4.  Everything else ([Q | delta] then [A | B | M | R] per stage) is unchanged. */
enum { SIP_LQR_LAYOUT_FULL = 0, SIP_LQR_LAYOUT_SYMMETRIC = 1 };

typedef struct sip_lqr_plan sip_lqr_plan;
/* Threading: like the reference (no globals; distinct LQR + Workspace pairs are independent, a
 * shared Workspace is not thread-safe), distinct plans may be used concurrently; one plan (and the
 * device buffers handed to it) serves one host thread / one stream at a time. */

/* Replaces: LQR::LQR(const Input&, Workspace&) + compile_topology()
 * (lqr.cpp:635-643) for `batch` problems of horizon T (= num_edges), state
 * dimension n, control dimension m, on HIP device `device`.  The chain
 * topology is compiled here once, like the reference caches its traversal
 * (lqr.cpp:641,646).  Every (dtype, n >= 1, m >= 1) is served: hot shapes by a
 * dedicated fused kernel, everything else by the general engine (one
 * wavefront per problem, see sip_lqr_kernel_name()). */
int sip_lqr_plan_create(int dtype, int64_t batch, int T, int n, int m,
                        int device, sip_lqr_plan **plan);
/* The same with a layout of `mats` other than FULL.  SIP_LQR_LAYOUT_SYMMETRIC: fp64 shapes with a symmetric-packed
 * fused kernel (sip_lqr_kernel_name() carries "sym": LDS-staged kernels with even n and m whose blocks stay whole
 * 16-byte pieces, the reference's n = 12, m = 4 benchmark shape among them); SIP_LQR_ERR_UNSUPPORTED otherwise.
 * sip_lqr_mats_len / _bytes, sip_lqr_split_mats_len, sip_lqr_pack_problem and every compute entry point follow the
 * plan's layout (sip_lqr_factor_solve_split: [Q | delta | M | R] with Q, R packed; what the Newton-KKT step's
 * condensation then writes); the one-sweep sip_lqr_solve_multi is FULL-layout only (a SYMMETRIC plan goes column by
 * column). */
int sip_lqr_plan_create_layout(int dtype, int64_t batch, int T, int n, int m, int device, int layout,
                               sip_lqr_plan **plan);
int sip_lqr_plan_layout(const sip_lqr_plan *plan);

void sip_lqr_plan_destroy(sip_lqr_plan *plan);

/* Sizes, in bytes, of the whole-batch buffers.  Replace
 * LQR::Workspace::num_bytes / LQR::Output::num_bytes (lqr.hpp:104-106,
 * 146-186); size_t, not int (the reference's int overflows at batch
 * scale). */
size_t sip_lqr_mats_bytes(const sip_lqr_plan *plan);
size_t sip_lqr_vecs_bytes(const sip_lqr_plan *plan);
size_t sip_lqr_sol_bytes(const sip_lqr_plan *plan);
size_t sip_lqr_gains_bytes(const sip_lqr_plan *plan);
size_t sip_lqr_status_bytes(const sip_lqr_plan *plan);
size_t sip_lqr_workspace_bytes(const sip_lqr_plan *plan);
/* Problems per call and bytes per scalar (8: SIP_LQR_F64, 4: SIP_LQR_F32) of a plan. */
int64_t sip_lqr_plan_batch(const sip_lqr_plan *plan);
size_t sip_lqr_scalar_bytes(const sip_lqr_plan *plan);
/* Per-problem lengths in scalars (packed chain layout above). */
size_t sip_lqr_mats_len(const sip_lqr_plan *plan);
size_t sip_lqr_vecs_len(const sip_lqr_plan *plan);
size_t sip_lqr_gains_len(const sip_lqr_plan *plan);

/* Host-side layout conversion between the reference's per-stage pointer
 * tables (LQR::Input, lqr.hpp:76-85: double** indexed by node or edge; the
 * blocks are always double, as in the reference) and the packed chain
 * layout of problem `p` inside host staging buffers of mats_bytes /
 * vecs_bytes / sol_bytes.  For SIP_LQR_F32 plans the values are rounded to
 * float on the way in and widened on the way out. */
int sip_lqr_pack_problem(const sip_lqr_plan *plan, int64_t p,
                         double *const *Q, double *const *M, double *const *R,
                         double *const *q, double *const *r, double *const *A,
                         double *const *B, double *const *c,
                         double *const *delta, void *mats_host,
                         void *vecs_host);
int sip_lqr_unpack_solution(const sip_lqr_plan *plan, int64_t p,
                            const void *sol_host, double *const *x,
                            double *const *u, double *const *y);
int sip_lqr_unpack_gains(const sip_lqr_plan *plan, int64_t p,
                         const void *gains_host, double *const *K,
                         double *const *k);

/* Replaces: the loop body of BM_LQRFactorSolve
 * (benchmarks/lqr_benchmark.cpp:653-663) = LQR::factor_with_status()
 * (lqr.cpp:645-731) followed by LQR::solve() (lqr.cpp:735-871), for every
 * problem of the batch, as one fused launch.  Writes status[p]; sol/gains of
 * a problem whose status != SUCCESS are unspecified (the reference's solve()
 * after a failed factor is undefined, lqr.cpp:735).  d_workspace:
 * sip_lqr_workspace_bytes() of device scratch holding the per-node factor
 * state the forward rollout needs (the role of LQR::Workspace::{W,V,
 * F_factor,v,...}, lqr.hpp:110-119). Asynchronous on `stream`. */
int sip_lqr_factor_solve(const sip_lqr_plan *plan, const void *d_mats,
                         const void *d_vecs, void *d_sol, void *d_gains,
                         int32_t *d_status, void *d_workspace, void *stream);

/* The same fused sweep with the dynamics Jacobians read in place.  Replaces: the copy of ddyn_dx /
 * ddyn_du into LQR::Input::A / B (helpers.cpp:365-366) followed by the loop body above -- the
 * Newton-KKT step (sip_kkt_factor_solve) hands the Riccati sweep the Jacobians where the model
 * callback left them instead of copying them into `mats`.
 *   d_mats: per problem sip_lqr_split_mats_len() scalars: node blocks [Q | delta] as in the packed
 *           chain layout, edge blocks [M | R] (no A, B);
 *   d_ab  : stage i of problem p: A (n x n, column-major) then B (n x m, column-major) at
 *           d_ab + (p * ab_problem_stride + i * ab_stage_stride) scalars.
 * d_mats, d_vecs, d_gains, d_workspace 16-byte aligned; d_ab and its strides any (8-byte aligned scalars: a stage's
 * A | B is a whole number of 16-byte pieces and is copied exactly from sources that are only 8-byte aligned);
 * 3 * ab_problem_stride * 8 + (n*n + n*m) * 8 < 2^32 bytes (the four
 * problems of a wavefront are addressed by 32-bit offsets; SIP_LQR_ERR_INVALID_ARGUMENT beyond: that is
 * a problem stride of up to ~178 M scalars).  Available (sip_lqr_has_split() == 1) for fp64 plans whose fused
 * kernel is an LDS-staged one (n <= 15, m <= 8) and whose A | B block is a whole number of 16-byte pieces
 * (n (n + m) even: 88 of those 120 shapes, the reference's Newton-KKT benchmark grid among them; with n (n + m)
 * odd the blocks of every other stage would start on an odd 8-byte offset); SIP_LQR_ERR_UNSUPPORTED otherwise.
 * Everything else as sip_lqr_factor_solve. */
int sip_lqr_has_split(const sip_lqr_plan *plan);
int64_t sip_lqr_split_mats_len(const sip_lqr_plan *plan);
int sip_lqr_factor_solve_split(const sip_lqr_plan *plan, const void *d_mats, const void *d_ab,
                               int64_t ab_problem_stride, int64_t ab_stage_stride, const void *d_vecs, void *d_sol,
                               void *d_gains, int32_t *d_status, void *d_workspace, void *stream);

/* Replaces: LQR::factor_with_status() alone (lqr.cpp:645-731; called by
 * CallbackProvider::factor, helpers.cpp:368).  Leaves the factor state in
 * d_workspace and the K part of d_gains for later sip_lqr_solve() calls.
 * Shapes with a fused kernel run its backward sweep alone (fp64: plus the LDL
 * factors of the G matrices for the vector-only solve mode); other shapes run
 * on the general engine, whose work arena holds the reference's factor state.
 * sol of a problem whose status != SUCCESS is unspecified. */
int sip_lqr_factor(const sip_lqr_plan *plan, const void *d_mats, void *d_gains,
                   int32_t *d_status, void *d_workspace, void *stream);

/* Replaces: LQR::solve(Output&) alone (lqr.cpp:735-871; called by
 * CallbackProvider::solve, helpers.cpp:826) against the factor state of the
 * last sip_lqr_factor() on the same workspace; may be called repeatedly with
 * new right-hand sides (tests/lqr_test.cpp:431-450).  d_gains: the buffer that
 * sip_lqr_factor() filled (K is read, the k part is written).  Shapes with a
 * fused fp64 kernel run the affine sweep alone (vectors distributed over the
 * lanes, no matrix work) followed by the rollout.  Fills sol. */
int sip_lqr_solve(const sip_lqr_plan *plan, const void *d_mats,
                  const void *d_vecs, void *d_sol, void *d_gains,
                  void *d_workspace, void *stream);

/* Replaces: the multi-right-hand-side block of solve_stagewise_kkt_matrix
 * (helpers.cpp:521-665: LQR::solve generalised from one right-hand side to
 * num_rhs columns, reading LQR::Workspace directly; the caller is the theta
 * Schur complement, helpers.cpp:387) against the factor state of the last
 * sip_lqr_factor() on the same workspace.  d_vecs_cols / d_sol_cols: num_rhs
 * arrays of sip_lqr_vecs_bytes() each, one after the other (column c at
 * c * sip_lqr_vecs_bytes()).  Shapes with a multi-rhs kernel (the fp64 shapes of
 * the reference's benchmark grid) carry up to 8 columns through ONE backward /
 * forward sweep, fetching every matrix operand of a stage once;
 * d_col_workspace: sip_lqr_solve_multi_workspace_bytes() of device scratch
 * (per-column g, h, k of the rollout).  Other shapes run sip_lqr_solve() per
 * column (workspace bytes 0, d_col_workspace may be NULL).  The k part of
 * d_gains is unspecified afterwards. */
size_t sip_lqr_solve_multi_workspace_bytes(const sip_lqr_plan *plan, int num_rhs);
int sip_lqr_solve_multi(const sip_lqr_plan *plan, const void *d_mats,
                        const void *d_vecs_cols, void *d_sol_cols, int num_rhs,
                        void *d_gains, void *d_workspace, void *d_col_workspace,
                        void *stream);

/* ------------------------------------------------------------------------
 * General trees / per-node dimensions (the full Topology + Dimensions model
 * of lqr.hpp:5-64): the path behind the drop-in C++ `LQR` adapter
 * (include/sip_optimal_control_amd/lqr_dropin.hpp).  `batch` instances of ONE
 * topology and dimension table, each with its own data; fp64; state and
 * control dimensions >= 0 (0 allowed, tests/variable_dimensions_test.cpp:
 * 316-336).
 *
 * Arenas (scalars = double, offsets from sip_lqr_tree_offset()):
 *   input  [batch][ node blocks: Q (n*n) | q (n) | c (n) | delta (n) ;
 *                   edge blocks: A (nc*np) | B (nc*m) | M (np*m) | R (m*m) | r (m) ]
 *   work   [batch][ edge blocks: W (max_n^2) | K (m*np) | G_factor (m*m) | k (m) ;
 *                   node blocks: V (n*n) | F_factor (n*n) | sqrt_delta (n) |
 *                                sqrt_delta_inv (n) | v (n) ; scratch ]
 *                  -- the fields of LQR::Workspace (lqr.hpp:110-127), same
 *                     meaning, so their consumers (helpers.cpp:521-665) can be
 *                     served from a copy of it
 *   output [batch][ node blocks: x (n) | y (n) ; edge blocks: u (m) ]
 * ------------------------------------------------------------------------ */
typedef struct sip_lqr_tree_plan sip_lqr_tree_plan;

/* Host-side topology compilation.  Replaces: compile_topology_data
 * (lqr.cpp:563-631): validates the edge list, builds the CSR children lists
 * (stable in edge index), DFS preorder (lowest edge index first) and the
 * reversed-preorder "postorder".  Output arrays are caller-owned with the sizes
 * of LQR::Workspace (lqr.hpp:129-135): child_offsets[N+1], child_edges[E],
 * edge_parents_out[E], edge_children_out[E], preorder[N], postorder[N],
 * node_marks[N].  Returns a FactorStatus (SUCCESS or INVALID_TOPOLOGY). */
int sip_lqr_compile_topology(int num_edges, int root, const int *edge_parents,
                             const int *edge_children, int *child_offsets,
                             int *child_edges, int *edge_parents_out,
                             int *edge_children_out, int *preorder,
                             int *postorder, int *node_marks);

/* Replaces: LQR::LQR + compile_topology (lqr.cpp:635-643) for `batch`
 * instances.  An invalid topology still yields a plan; its status is latched
 * (lqr.cpp:646-648) and returned by every sip_lqr_tree_factor(). */
int sip_lqr_tree_plan_create(int64_t batch, int num_edges, int root,
                             const int *edge_parents, const int *edge_children,
                             const int *state_dims, const int *control_dims,
                             int device, sip_lqr_tree_plan **plan);
void sip_lqr_tree_plan_destroy(sip_lqr_tree_plan *plan);
int sip_lqr_tree_topology_status(const sip_lqr_tree_plan *plan);
/* Compiled traversal (host copies, valid when the topology is valid):
 * which = 0 child_offsets[N+1], 1 child_edges[E], 2 preorder[N], 3 postorder[N] */
const int *sip_lqr_tree_topology_array(const sip_lqr_tree_plan *plan, int which);

/* Per-problem arena lengths in doubles. */
size_t sip_lqr_tree_input_len(const sip_lqr_tree_plan *plan);
size_t sip_lqr_tree_work_len(const sip_lqr_tree_plan *plan);
size_t sip_lqr_tree_output_len(const sip_lqr_tree_plan *plan);
/* Offset (in doubles, inside one problem) of a block: arena 0 = input,
 * 1 = work, 2 = output; kind 0 = node block, 1 = edge block; index = node or
 * edge id.  Returns (size_t)-1 on a bad argument. */
size_t sip_lqr_tree_offset(const sip_lqr_tree_plan *plan, int arena, int kind,
                           int index);

/* Replaces: LQR::factor_with_status() (lqr.cpp:645-731) for every instance:
 * reads Q, M, R, A, B, delta from d_input, writes the factor state to d_work
 * and FactorStatus codes to d_status[batch].  Asynchronous on `stream`. */
int sip_lqr_tree_factor(const sip_lqr_tree_plan *plan, const double *d_input,
                        double *d_work, int32_t *d_status, void *stream);
/* Replaces: LQR::solve(Output&) (lqr.cpp:735-871): reads q, r, c (and A, B,
 * delta) from d_input and the factor state from d_work, writes x, u, y to
 * d_output and k, v to d_work.  Instances whose d_status != SUCCESS are
 * skipped.  May be called repeatedly after one factor. */
int sip_lqr_tree_solve(const sip_lqr_tree_plan *plan, const double *d_input,
                       double *d_work, double *d_output,
                       const int32_t *d_status, void *stream);

/* Replaces: factor_with_status() + solve() (the loop body of BM_LQRVariableFactorSolve,
 * benchmarks/lqr_benchmark.cpp:716-744) for every instance, as one fused sweep: trees whose state
 * dimensions are <= 15 and control dimensions <= 8 run on a padded size class of the
 * broadcast-FMA kernels (csrc/tree_qw16.hpp; sip_lqr_tree_kernel_name() tells), anything else on the
 * general engine (factor, then solve).  Fills d_output (x, y, u), d_status and, of d_work, the
 * K and k fields of every edge (d_work may be NULL on the fused path: no gains wanted); the other
 * LQR::Workspace fields of d_work are only filled by sip_lqr_tree_factor().  d_scratch:
 * sip_lqr_tree_fused_scratch_bytes() bytes of device scratch (0: general engine, may be NULL). */
size_t sip_lqr_tree_fused_scratch_bytes(const sip_lqr_tree_plan *plan);
int sip_lqr_tree_factor_solve(const sip_lqr_tree_plan *plan, const double *d_input,
                              double *d_work, double *d_output, int32_t *d_status,
                              void *d_scratch, void *stream);
/* The same fused sweep, leaving in d_work (required) EVERY factor-state field of LQR::Workspace that
 * sip_lqr_tree_factor() leaves there (lqr.hpp:109-135: W, K, G_factor, k per edge; V, F_factor, sqrt_delta,
 * sqrt_delta_inv, v per node -- read by helpers.cpp:521-665), same arena layout: what the drop-in LQR class needs to
 * run on the fused kernels by default.  G_factor / F_factor: the lower triangle holds Eigen's L, the entries above
 * the diagonal their pre-factor values (Eigen::LLT factors in place, lqr.cpp:505, 697).  A second instantiation of
 * the kernel, so sip_lqr_tree_factor_solve() keeps its register budget. */
int sip_lqr_tree_factor_solve_workspace(const sip_lqr_tree_plan *plan, const double *d_input,
                                        double *d_work, double *d_output, int32_t *d_status,
                                        void *d_scratch, void *stream);
const char *sip_lqr_tree_kernel_name(const sip_lqr_tree_plan *plan);

/* Name of the kernel variant the plan dispatches to (static string). */
const char *sip_lqr_kernel_name(const sip_lqr_plan *plan);

/* Library version string. */
const char *sip_lqr_version(void);

#ifdef __cplusplus
}
#endif
#endif
