// kkt_kernels.hpp -- device side of the batched Newton-KKT step
// (include/sip_kkt_amd.h): the work either side of the Riccati solve.
//
//   weights_kernel   1/r2, 1/(w+r3), positivity check   helpers.cpp:251-297
//   condense_kernel  Q_mod, M_mod, R_mod, A, B, dyn_r2   helpers.cpp:299-367
//   rhs_kernel       q_mod, r_mod, c_mod from b          helpers.cpp:752-812
//   recover_kernel   x,u,y scatter + multipliers y_c, z  helpers.cpp:817-892
//   apply_kernel     y += K x                            helpers.cpp:953-1368
//
// All of it is HBM-bound elementwise / rank-k work on small blocks: one
// 64-lane workgroup per (problem, node-or-edge) item, lanes over the output
// elements of the item, every output element accumulated by exactly one lane
// in the reference's order (constraint-major, child edges in index order), so
// there are no atomics and the result does not depend on scheduling.  A node
// item gathers the contributions of its child edges (the reference scatters
// them from the edge loop into the parent's block).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sipamd {
namespace kkt {

constexpr int TPB = 64;

enum Block {
  N_Q = 0, N_JC, N_JG, E_Q, E_M, E_R, E_A, E_B, E_JXC, E_JUC, E_JXG, E_JUG, NUM_BLOCKS
};

struct Meta {
  int E, N, root;
  int x_dim, y_dim, z_dim; // stagewise (theta excluded)
  int theta_dim;           // 0 except for apply_kernel on [x | theta | y | z] vectors
  long model_len, in0_len, in1_len, out_len;
  // dimension tables and traversal (per node / per edge)
  const int *sd, *cd, *ncd, *ngd, *ecd, *egd, *parent, *child, *in_edge, *child_offsets, *child_edges;
  // flattened KKT vector offsets, types.cpp:33-63
  const int *x_state, *x_control, *y_dyn, *y_node_c, *y_edge_c, *z_node, *z_edge;
  // 1 where a y row is a dynamics row (weight = r2 itself), else 0
  const int *y_is_dyn;
  const long *mo[NUM_BLOCKS]; // model arena block offsets
  // LDS plan of the staged kernels (doubles): Q accumulator | one model item |
  // the weighted copy of its Jacobian tail | weighted right-hand side rows
  int lds_q, lds_item, lds_tail, lds_rows;
  // offsets inside the LQR arenas (generic_plan.hpp tables)
  const long *oQ, *od, *oq, *oc, *ox, *oy, *oA, *oB, *oM, *oR, *orr, *ou;
};

// inv[0 .. y_dim) : dyn rows r2, constraint rows 1/r2; inv[y_dim ..) : 1/(w+r3).
// regstat[p] != 0 iff any regularization of problem p is <= 0.
__global__ void __launch_bounds__(256)
weights_kernel(const Meta mt, const double *__restrict__ w, const double *__restrict__ r2,
               const double *__restrict__ r3, double *__restrict__ inv, int *__restrict__ regstat,
               long batch) {
  const long per = (long)mt.y_dim + mt.z_dim;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= batch * per)
    return;
  const long p = idx / per;
  const int k = (int)(idx - p * per);
  double reg, out;
  if (k < mt.y_dim) {
    reg = r2[p * mt.y_dim + k];
    out = mt.y_is_dyn[k] ? reg : 1.0 / reg;
  } else {
    const long at = p * mt.z_dim + (k - mt.y_dim);
    reg = w[at] + r3[at];
    out = 1.0 / reg;
  }
  inv[idx] = out;
  if (reg <= 0.0)
    atomicOr(&regstat[p], 1);
}

__global__ void __launch_bounds__(256)
merge_status_kernel(const int *__restrict__ regstat, int32_t *__restrict__ status, long batch, int code) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < batch && regstat[p] != 0)
    status[p] = code;
}

// acc += sum_k (wgt[k] * Jb[k, cb]) * Ja[k, ca]   (J column-major, `rows` rows)
__device__ __forceinline__ double rank_update(double acc, const double *Ja, int ca, const double *Jb, int cb,
                                              int rows, const double *wgt) {
  for (int k = 0; k < rows; ++k)
    acc += (wgt[k] * Jb[k + (long)rows * cb]) * Ja[k + (long)rows * ca];
  return acc;
}

__global__ void __launch_bounds__(TPB)
condense_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ r1_all,
                const double *__restrict__ inv_all, double *__restrict__ in0_all, long batch) {
  const int items = mt.N + mt.E;
  const long p = blockIdx.x / items;
  const int item = blockIdx.x - (unsigned)(p * items);
  if (p >= batch)
    return;
  const double *model = model_all + p * mt.model_len;
  const double *r1 = r1_all + p * mt.x_dim;
  const double *yinv = inv_all + p * ((long)mt.y_dim + mt.z_dim), *zinv = yinv + mt.y_dim;
  double *in0 = in0_all + p * mt.in0_len;
  const int tid = threadIdx.x;
  if (item < mt.N) {
    const int i = item, n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];
    const double *Q = model + mt.mo[N_Q][i], *Jc = model + mt.mo[N_JC][i], *Jg = model + mt.mo[N_JG][i];
    double *Qm = in0 + mt.oQ[i], *dl = in0 + mt.od[i];
    for (int d = tid; d < n; d += TPB)
      dl[d] = yinv[mt.y_dyn[i] + d]; // dyn_r2, helpers.cpp:255-261
    const int lo = mt.child_offsets[i], hi = mt.child_offsets[i + 1];
    for (int idx = tid; idx < n * n; idx += TPB) {
      const int col = idx / n, row = idx - col * n;
      if (row < col)
        continue;
      double acc = Q[row + (long)n * col]; // lower triangle of d2L_dx2, :310-311
      if (row == col)
        acc += r1[mt.x_state[i] + row]; // :312-313
      acc = rank_update(acc, Jc, row, Jc, col, c, yinv + mt.y_node_c[i]); // :314
      acc = rank_update(acc, Jg, row, Jg, col, g, zinv + mt.z_node[i]);   // :315
      for (int ci = lo; ci < hi; ++ci) { // the edge loop's additions to Q_mod[parent], :336-339
        const int e = mt.child_edges[ci];
        acc += (model + mt.mo[E_Q][e])[row + (long)n * col];
        const double *Jxc = model + mt.mo[E_JXC][e], *Jxg = model + mt.mo[E_JXG][e];
        acc = rank_update(acc, Jxc, row, Jxc, col, mt.ecd[e], yinv + mt.y_edge_c[e]);
        acc = rank_update(acc, Jxg, row, Jxg, col, mt.egd[e], zinv + mt.z_edge[e]);
      }
      Qm[row + (long)n * col] = acc;
      Qm[col + (long)n * row] = acc; // mirror_lower_to_upper, :357-361
    }
  } else {
    const int e = item - mt.N, pa = mt.parent[e], n = mt.sd[pa], nc = mt.sd[mt.child[e]], m = mt.cd[e];
    const int c = mt.ecd[e], g = mt.egd[e];
    const double *Jxc = model + mt.mo[E_JXC][e], *Juc = model + mt.mo[E_JUC][e];
    const double *Jxg = model + mt.mo[E_JXG][e], *Jug = model + mt.mo[E_JUG][e];
    const double *wc = yinv + mt.y_edge_c[e], *wg = zinv + mt.z_edge[e];
    const double *M = model + mt.mo[E_M][e], *R = model + mt.mo[E_R][e];
    double *Mm = in0 + mt.oM[e], *Rm = in0 + mt.oR[e];
    for (int idx = tid; idx < n * m; idx += TPB) { // M_mod, :341-343, 349-352
      const int col = idx / n, row = idx - col * n;
      double acc = M[idx];
      acc = rank_update(acc, Jxc, row, Juc, col, c, wc);
      acc = rank_update(acc, Jxg, row, Jug, col, g, wg);
      Mm[idx] = acc;
    }
    for (int idx = tid; idx < m * m; idx += TPB) { // R_mod, :344-353
      const int col = idx / m, row = idx - col * m;
      if (row < col)
        continue;
      double acc = R[row + (long)m * col];
      if (row == col)
        acc += r1[mt.x_control[e] + row];
      acc = rank_update(acc, Juc, row, Juc, col, c, wc);
      acc = rank_update(acc, Jug, row, Jug, col, g, wg);
      Rm[row + (long)m * col] = acc;
      Rm[col + (long)m * row] = acc;
    }
    const double *A = model + mt.mo[E_A][e], *B = model + mt.mo[E_B][e]; // :365-366
    double *Am = in0 + mt.oA[e], *Bm = in0 + mt.oB[e];
    for (int idx = tid; idx < nc * n; idx += TPB)
      Am[idx] = A[idx];
    for (int idx = tid; idx < nc * m; idx += TPB)
      Bm[idx] = B[idx];
  }
}

// acc -= sum_k J[k, col] * (wgt[k] * rhs[k])   (subtract_weighted_jacobian_rhs, :138-153)
__device__ __forceinline__ double sub_weighted(double acc, const double *J, int col, int rows,
                                               const double *wgt, const double *rhs) {
  for (int k = 0; k < rows; ++k)
    acc -= J[k + (long)rows * col] * (wgt[k] * rhs[k]);
  return acc;
}

__global__ void __launch_bounds__(TPB)
rhs_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ b_all,
           const double *__restrict__ inv_all, double *__restrict__ in1_all, const int32_t *__restrict__ status,
           long batch) {
  const int items = mt.N + mt.E;
  const long p = blockIdx.x / items;
  const int item = blockIdx.x - (unsigned)(p * items);
  if (p >= batch || (status != nullptr && status[p] != 0))
    return;
  const long kkt = (long)mt.x_dim + mt.y_dim + mt.z_dim;
  const double *model = model_all + p * mt.model_len;
  const double *b = b_all + p * kkt, *b_y = b + mt.x_dim, *b_z = b_y + mt.y_dim;
  const double *yinv = inv_all + p * ((long)mt.y_dim + mt.z_dim), *zinv = yinv + mt.y_dim;
  double *in1 = in1_all + p * mt.in1_len;
  const int tid = threadIdx.x;
  if (item < mt.N) {
    const int i = item, n = mt.sd[i];
    const int lo = mt.child_offsets[i], hi = mt.child_offsets[i + 1];
    for (int d = tid; d < n; d += TPB) {
      double acc = -b[mt.x_state[i] + d]; // :772
      acc = sub_weighted(acc, model + mt.mo[N_JC][i], d, mt.ncd[i], yinv + mt.y_node_c[i], b_y + mt.y_node_c[i]);
      acc = sub_weighted(acc, model + mt.mo[N_JG][i], d, mt.ngd[i], zinv + mt.z_node[i], b_z + mt.z_node[i]);
      for (int ci = lo; ci < hi; ++ci) { // :805-806
        const int e = mt.child_edges[ci];
        acc = sub_weighted(acc, model + mt.mo[E_JXC][e], d, mt.ecd[e], yinv + mt.y_edge_c[e], b_y + mt.y_edge_c[e]);
        acc = sub_weighted(acc, model + mt.mo[E_JXG][e], d, mt.egd[e], zinv + mt.z_edge[e], b_z + mt.z_edge[e]);
      }
      in1[mt.oq[i] + d] = acc;
      in1[mt.oc[i] + d] = -b_y[mt.y_dyn[i] + d]; // c_mod, :776-777
    }
  } else {
    const int e = item - mt.N, m = mt.cd[e];
    for (int d = tid; d < m; d += TPB) {
      double acc = -b[mt.x_control[e] + d]; // :807
      acc = sub_weighted(acc, model + mt.mo[E_JUC][e], d, mt.ecd[e], yinv + mt.y_edge_c[e], b_y + mt.y_edge_c[e]);
      acc = sub_weighted(acc, model + mt.mo[E_JUG][e], d, mt.egd[e], zinv + mt.z_edge[e], b_z + mt.z_edge[e]);
      in1[mt.orr[e] + d] = acc;
    }
  }
}

// row k of J x  (J column-major rows x cols)
__device__ __forceinline__ double row_dot(const double *J, int k, int rows, int cols, const double *x) {
  double acc = 0.0;
  for (int col = 0; col < cols; ++col)
    acc += J[k + (long)rows * col] * x[col];
  return acc;
}

__global__ void __launch_bounds__(TPB)
recover_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ b_all,
               const double *__restrict__ inv_all, const double *__restrict__ out_all,
               double *__restrict__ sol_all, const int32_t *__restrict__ status, long batch) {
  const int items = mt.N + mt.E;
  const long p = blockIdx.x / items;
  const int item = blockIdx.x - (unsigned)(p * items);
  if (p >= batch || status[p] != 0)
    return;
  const long kkt = (long)mt.x_dim + mt.y_dim + mt.z_dim;
  const double *model = model_all + p * mt.model_len;
  const double *b_y = b_all + p * kkt + mt.x_dim, *b_z = b_y + mt.y_dim;
  const double *yinv = inv_all + p * ((long)mt.y_dim + mt.z_dim), *zinv = yinv + mt.y_dim;
  const double *out = out_all + p * mt.out_len;
  double *sol = sol_all + p * kkt, *sol_y = sol + mt.x_dim, *sol_z = sol_y + mt.y_dim;
  const int tid = threadIdx.x;
  if (item < mt.N) {
    const int i = item, n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];
    const double *x = out + mt.ox[i];
    for (int d = tid; d < n; d += TPB) { // the aliasing of :817-824
      sol[mt.x_state[i] + d] = x[d];
      sol_y[mt.y_dyn[i] + d] = out[mt.oy[i] + d];
    }
    for (int k = tid; k < c + g; k += TPB) { // :828-855
      if (k < c) {
        const int at = mt.y_node_c[i] + k;
        sol_y[at] = (row_dot(model + mt.mo[N_JC][i], k, c, n, x) - b_y[at]) * yinv[at];
      } else {
        const int at = mt.z_node[i] + (k - c);
        sol_z[at] = (row_dot(model + mt.mo[N_JG][i], k - c, g, n, x) - b_z[at]) * zinv[at];
      }
    }
  } else {
    const int e = item - mt.N, pa = mt.parent[e], n = mt.sd[pa], m = mt.cd[e], c = mt.ecd[e], g = mt.egd[e];
    const double *x = out + mt.ox[pa], *u = out + mt.ou[e];
    for (int d = tid; d < m; d += TPB)
      sol[mt.x_control[e] + d] = u[d];
    for (int k = tid; k < c + g; k += TPB) { // :857-892
      if (k < c) {
        const int at = mt.y_edge_c[e] + k;
        const double jx = row_dot(model + mt.mo[E_JXC][e], k, c, n, x);
        const double ju = row_dot(model + mt.mo[E_JUC][e], k, c, m, u);
        sol_y[at] = ((jx + ju) - b_y[at]) * yinv[at];
      } else {
        const int at = mt.z_edge[e] + (k - c);
        const double jx = row_dot(model + mt.mo[E_JXG][e], k - c, g, n, x);
        const double ju = row_dot(model + mt.mo[E_JUG][e], k - c, g, m, u);
        sol_z[at] = ((jx + ju) - b_z[at]) * zinv[at];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Staged variants: one workgroup per (problem, node).  The model item of the
// node ([d2L_dx2 | dc_dx | dg_dx], contiguous) and then of each child edge
// ([d2L_dx2 | d2L_dxdu | d2L_du2 | ddyn_dx | ddyn_du | dc_dx | dc_du | dg_dx |
// dg_du], contiguous) is copied HBM -> LDS with coalesced loads, the
// weighted Jacobians w o J are formed once in LDS, and every output element is
// accumulated from LDS by one lane in the same order and with the same
// operations as the direct kernels above (bit-identical results).  Q_mod is
// accumulated in LDS across the child edges; q_mod in a register.
// ---------------------------------------------------------------------------
// All the loads of a pass are issued before the first LDS store (a plain copy
// loop waits for each load in turn: one HBM round trip per 512 bytes).
__device__ __forceinline__ void stage_copy(double *dst, const double *__restrict__ src, int len, int tid) {
  constexpr int U = 8;
  for (int base = tid; base < len; base += U * TPB) {
    double v[U];
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

// acc += sum_k JW[k, cb] * J[k, ca]   (both column-major with `rows` rows, in LDS)
__device__ __forceinline__ double rank_update_w(double acc, const double *J, int ca, const double *JW, int cb,
                                                int rows) {
  const double *a = J + rows * ca, *b = JW + rows * cb;
  for (int k = 0; k < rows; ++k)
    acc += b[k] * a[k];
  return acc;
}

// acc -= sum_k J[k, col] * wr[k]
__device__ __forceinline__ double sub_weighted_w(double acc, const double *J, int col, int rows, const double *wr) {
  const double *a = J + rows * col;
  for (int k = 0; k < rows; ++k)
    acc -= a[k] * wr[k];
  return acc;
}

template <bool WITH_RHS>
__global__ void __launch_bounds__(TPB)
condense_staged_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ r1_all,
                       const double *__restrict__ inv_all, double *__restrict__ in0_all,
                       const double *__restrict__ b_all, double *__restrict__ in1_all, long batch) {
  extern __shared__ double sm[];
  double *Qacc = sm, *buf = Qacc + mt.lds_q, *jw = buf + mt.lds_item, *wr = jw + mt.lds_tail;
  const long p = blockIdx.x / mt.N;
  const int i = blockIdx.x - (unsigned)(p * mt.N);
  if (p >= batch)
    return;
  const long kkt = (long)mt.x_dim + mt.y_dim + mt.z_dim;
  const double *model = model_all + p * mt.model_len;
  const double *r1 = r1_all + p * mt.x_dim;
  const double *yinv = inv_all + p * ((long)mt.y_dim + mt.z_dim), *zinv = yinv + mt.y_dim;
  const double *b = WITH_RHS ? b_all + p * kkt : nullptr;
  const double *b_y = WITH_RHS ? b + mt.x_dim : nullptr, *b_z = WITH_RHS ? b_y + mt.y_dim : nullptr;
  double *in0 = in0_all + p * mt.in0_len;
  double *in1 = WITH_RHS ? in1_all + p * mt.in1_len : nullptr;
  const int tid = threadIdx.x;
  const int n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];

  // ---- the node's own terms (helpers.cpp:299-318, 752-779) ----
  stage_copy(buf, model + mt.mo[N_Q][i], n * n + (c + g) * n, tid);
  __syncthreads();
  {
    const double *Jc = buf + n * n, *Jg = Jc + c * n;
    const double *wc = yinv + mt.y_node_c[i], *wg = zinv + mt.z_node[i];
    for (int idx = tid; idx < c * n; idx += TPB)
      jw[idx] = wc[idx % c] * Jc[idx];
    for (int idx = tid; idx < g * n; idx += TPB)
      jw[c * n + idx] = wg[idx % g] * Jg[idx];
    if (WITH_RHS) {
      for (int k = tid; k < c + g; k += TPB)
        wr[k] = k < c ? wc[k] * b_y[mt.y_node_c[i] + k] : wg[k - c] * b_z[mt.z_node[i] + (k - c)];
    }
  }
  __syncthreads();
  double qacc = 0.0; // lane d < n: q_mod[d] (n <= TPB handled by the loop below otherwise)
  {
    const double *Jc = buf + n * n, *Jg = Jc + c * n;
    for (int idx = tid; idx < n * n; idx += TPB) {
      const int col = idx / n, row = idx - col * n;
      if (row < col)
        continue;
      double acc = buf[idx];
      if (row == col)
        acc += r1[mt.x_state[i] + row];
      acc = rank_update_w(acc, Jc, row, jw, col, c);
      acc = rank_update_w(acc, Jg, row, jw + c * n, col, g);
      Qacc[idx] = acc;
    }
    if (WITH_RHS) {
      for (int d = tid; d < n; d += TPB) {
        double acc = -b[mt.x_state[i] + d];
        acc = sub_weighted_w(acc, Jc, d, c, wr);
        acc = sub_weighted_w(acc, Jg, d, g, wr + c);
        if (n <= TPB)
          qacc = acc;
        else
          in1[mt.oq[i] + d] = acc;
        in1[mt.oc[i] + d] = -b_y[mt.y_dyn[i] + d];
      }
    }
    for (int d = tid; d < n; d += TPB)
      in0[mt.od[i] + d] = yinv[mt.y_dyn[i] + d];
  }

  // ---- child edges in index order (helpers.cpp:320-355, 781-812) ----
  for (int ci = mt.child_offsets[i]; ci < mt.child_offsets[i + 1]; ++ci) {
    const int e = mt.child_edges[ci];
    const int nc = mt.sd[mt.child[e]], m = mt.cd[e], ce = mt.ecd[e], ge = mt.egd[e];
    const int o_m = n * n, o_r = o_m + n * m, o_a = o_r + m * m, o_b = o_a + nc * n, o_j = o_b + nc * m;
    const int tail = (ce + ge) * (n + m);
    __syncthreads(); // previous consumers of buf / jw / wr are done
    stage_copy(buf, model + mt.mo[E_Q][e], o_j + tail, tid);
    __syncthreads();
    const double *Jxc = buf + o_j, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
    double *wxc = jw, *wuc = wxc + ce * n, *wxg = wuc + ce * m, *wug = wxg + ge * n;
    {
      const double *wc = yinv + mt.y_edge_c[e], *wg = zinv + mt.z_edge[e];
      for (int idx = tid; idx < ce * (n + m); idx += TPB)
        wxc[idx] = wc[idx % ce] * Jxc[idx]; // [Jxc | Juc] share the row count
      for (int idx = tid; idx < ge * (n + m); idx += TPB)
        wxg[idx] = wg[idx % ge] * Jxg[idx];
      if (WITH_RHS) {
        for (int k = tid; k < ce + ge; k += TPB)
          wr[k] = k < ce ? wc[k] * b_y[mt.y_edge_c[e] + k] : wg[k - ce] * b_z[mt.z_edge[e] + (k - ce)];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += TPB) { // Q_mod[parent] += ..., :336-339
      const int col = idx / n, row = idx - col * n;
      if (row < col)
        continue;
      double acc = Qacc[idx] + buf[idx];
      acc = rank_update_w(acc, Jxc, row, wxc, col, ce);
      acc = rank_update_w(acc, Jxg, row, wxg, col, ge);
      Qacc[idx] = acc;
    }
    double *Mm = in0 + mt.oM[e], *Rm = in0 + mt.oR[e];
    for (int idx = tid; idx < n * m; idx += TPB) { // M_mod, :341-352
      const int col = idx / n, row = idx - col * n;
      double acc = buf[o_m + idx];
      acc = rank_update_w(acc, Jxc, row, wuc, col, ce);
      acc = rank_update_w(acc, Jxg, row, wug, col, ge);
      Mm[idx] = acc;
    }
    for (int idx = tid; idx < m * m; idx += TPB) { // R_mod, :344-353
      const int col = idx / m, row = idx - col * m;
      if (row < col)
        continue;
      double acc = buf[o_r + row + m * col];
      if (row == col)
        acc += r1[mt.x_control[e] + row];
      acc = rank_update_w(acc, Juc, row, wuc, col, ce);
      acc = rank_update_w(acc, Jug, row, wug, col, ge);
      Rm[row + (long)m * col] = acc;
      Rm[col + (long)m * row] = acc;
    }
    double *Am = in0 + mt.oA[e], *Bm = in0 + mt.oB[e]; // :365-366
    for (int idx = tid; idx < nc * n; idx += TPB)
      Am[idx] = buf[o_a + idx];
    for (int idx = tid; idx < nc * m; idx += TPB)
      Bm[idx] = buf[o_b + idx];
    if (WITH_RHS) {
      for (int d = tid; d < n; d += TPB) { // q_mod[parent] -= ..., :805-806
        double acc = n <= TPB ? qacc : in1[mt.oq[i] + d];
        acc = sub_weighted_w(acc, Jxc, d, ce, wr);
        acc = sub_weighted_w(acc, Jxg, d, ge, wr + ce);
        if (n <= TPB)
          qacc = acc;
        else
          in1[mt.oq[i] + d] = acc;
      }
      for (int d = tid; d < m; d += TPB) { // r_mod, :807-809
        double acc = -b[mt.x_control[e] + d];
        acc = sub_weighted_w(acc, Juc, d, ce, wr);
        acc = sub_weighted_w(acc, Jug, d, ge, wr + ce);
        in1[mt.orr[e] + d] = acc;
      }
    }
  }
  if (WITH_RHS && n <= TPB && tid < n)
    in1[mt.oq[i] + tid] = qacc;
  // Q_mod out, mirrored (:357-361); each lane writes the elements it accumulated
  double *Qm = in0 + mt.oQ[i];
  for (int idx = tid; idx < n * n; idx += TPB) {
    const int col = idx / n, row = idx - col * n;
    if (row < col)
      continue;
    const double acc = Qacc[idx];
    Qm[idx] = acc;
    Qm[col + (long)n * row] = acc;
  }
}

// rhs alone (the split solve path): stages only the Jacobian tails.
__global__ void __launch_bounds__(TPB)
rhs_staged_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ b_all,
                  const double *__restrict__ inv_all, double *__restrict__ in1_all,
                  const int32_t *__restrict__ status, long batch) {
  extern __shared__ double sm[];
  double *buf = sm, *wr = buf + mt.lds_tail;
  const long p = blockIdx.x / mt.N;
  const int i = blockIdx.x - (unsigned)(p * mt.N);
  if (p >= batch || (status != nullptr && status[p] != 0))
    return;
  const long kkt = (long)mt.x_dim + mt.y_dim + mt.z_dim;
  const double *model = model_all + p * mt.model_len;
  const double *b = b_all + p * kkt, *b_y = b + mt.x_dim, *b_z = b_y + mt.y_dim;
  const double *yinv = inv_all + p * ((long)mt.y_dim + mt.z_dim), *zinv = yinv + mt.y_dim;
  double *in1 = in1_all + p * mt.in1_len;
  const int tid = threadIdx.x;
  const int n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];
  stage_copy(buf, model + mt.mo[N_JC][i], (c + g) * n, tid);
  for (int k = tid; k < c + g; k += TPB)
    wr[k] = k < c ? yinv[mt.y_node_c[i] + k] * b_y[mt.y_node_c[i] + k]
                  : zinv[mt.z_node[i] + (k - c)] * b_z[mt.z_node[i] + (k - c)];
  __syncthreads();
  for (int d = tid; d < n; d += TPB) {
    double acc = -b[mt.x_state[i] + d];
    acc = sub_weighted_w(acc, buf, d, c, wr);
    acc = sub_weighted_w(acc, buf + c * n, d, g, wr + c);
    in1[mt.oq[i] + d] = acc;
    in1[mt.oc[i] + d] = -b_y[mt.y_dyn[i] + d];
  }
  for (int ci = mt.child_offsets[i]; ci < mt.child_offsets[i + 1]; ++ci) {
    const int e = mt.child_edges[ci], m = mt.cd[e], ce = mt.ecd[e], ge = mt.egd[e];
    __syncthreads();
    stage_copy(buf, model + mt.mo[E_JXC][e], (ce + ge) * (n + m), tid);
    for (int k = tid; k < ce + ge; k += TPB)
      wr[k] = k < ce ? yinv[mt.y_edge_c[e] + k] * b_y[mt.y_edge_c[e] + k]
                     : zinv[mt.z_edge[e] + (k - ce)] * b_z[mt.z_edge[e] + (k - ce)];
    __syncthreads();
    const double *Jxc = buf, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
    for (int d = tid; d < n; d += TPB) {
      double acc = in1[mt.oq[i] + d]; // written by this lane above
      acc = sub_weighted_w(acc, Jxc, d, ce, wr);
      acc = sub_weighted_w(acc, Jxg, d, ge, wr + ce);
      in1[mt.oq[i] + d] = acc;
    }
    for (int d = tid; d < m; d += TPB) {
      double acc = -b[mt.x_control[e] + d];
      acc = sub_weighted_w(acc, Juc, d, ce, wr);
      acc = sub_weighted_w(acc, Jug, d, ge, wr + ce);
      in1[mt.orr[e] + d] = acc;
    }
  }
}

// recover: stages the Jacobian tails and the node's x (and each edge's u).
__global__ void __launch_bounds__(TPB)
recover_staged_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ b_all,
                      const double *__restrict__ inv_all, const double *__restrict__ out_all,
                      double *__restrict__ sol_all, const int32_t *__restrict__ status, long batch) {
  extern __shared__ double sm[];
  double *buf = sm, *xs = buf + mt.lds_tail, *us = xs + mt.lds_rows; // lds_rows >= max n, max m
  const long p = blockIdx.x / mt.N;
  const int i = blockIdx.x - (unsigned)(p * mt.N);
  if (p >= batch || status[p] != 0)
    return;
  const long kkt = (long)mt.x_dim + mt.y_dim + mt.z_dim;
  const double *model = model_all + p * mt.model_len;
  const double *b_y = b_all + p * kkt + mt.x_dim, *b_z = b_y + mt.y_dim;
  const double *yinv = inv_all + p * ((long)mt.y_dim + mt.z_dim), *zinv = yinv + mt.y_dim;
  const double *out = out_all + p * mt.out_len;
  double *sol = sol_all + p * kkt, *sol_y = sol + mt.x_dim, *sol_z = sol_y + mt.y_dim;
  const int tid = threadIdx.x;
  const int n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];
  stage_copy(buf, model + mt.mo[N_JC][i], (c + g) * n, tid);
  for (int d = tid; d < n; d += TPB) {
    const double xv = out[mt.ox[i] + d];
    xs[d] = xv;
    sol[mt.x_state[i] + d] = xv;
    sol_y[mt.y_dyn[i] + d] = out[mt.oy[i] + d];
  }
  __syncthreads();
  for (int k = tid; k < c + g; k += TPB) {
    if (k < c) {
      const int at = mt.y_node_c[i] + k;
      sol_y[at] = (row_dot(buf, k, c, n, xs) - b_y[at]) * yinv[at];
    } else {
      const int at = mt.z_node[i] + (k - c);
      sol_z[at] = (row_dot(buf + c * n, k - c, g, n, xs) - b_z[at]) * zinv[at];
    }
  }
  for (int ci = mt.child_offsets[i]; ci < mt.child_offsets[i + 1]; ++ci) {
    const int e = mt.child_edges[ci], m = mt.cd[e], ce = mt.ecd[e], ge = mt.egd[e];
    __syncthreads();
    stage_copy(buf, model + mt.mo[E_JXC][e], (ce + ge) * (n + m), tid);
    for (int d = tid; d < m; d += TPB) {
      const double uv = out[mt.ou[e] + d];
      us[d] = uv;
      sol[mt.x_control[e] + d] = uv;
    }
    __syncthreads();
    const double *Jxc = buf, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
    for (int k = tid; k < ce + ge; k += TPB) {
      if (k < ce) {
        const int at = mt.y_edge_c[e] + k;
        const double jx = row_dot(Jxc, k, ce, n, xs), ju = row_dot(Juc, k, ce, m, us);
        sol_y[at] = ((jx + ju) - b_y[at]) * yinv[at];
      } else {
        const int at = mt.z_edge[e] + (k - ce);
        const double jx = row_dot(Jxg, k - ce, ge, n, xs), ju = row_dot(Jug, k - ce, ge, m, us);
        sol_z[at] = ((jx + ju) - b_z[at]) * zinv[at];
      }
    }
  }
}

// column `col` of J^T v : sum_k J[k, col] v[k]
__device__ __forceinline__ double col_dot(const double *J, int col, int rows, const double *v) {
  double acc = 0.0;
  for (int k = 0; k < rows; ++k)
    acc += J[k + (long)rows * col] * v[k];
  return acc;
}

// Which blocks of K = [[H + r1, C^T, G^T], [C, -r2, 0], [G, 0, -(w + r3)]] an apply launch adds:
// CallbackProvider::add_Kx_to_y (helpers.cpp:953-976) is all of them, the five operators SIP is
// handed one by one (helpers.hpp:20-24; sip_optimal_control.cpp:147-190) are one bit each.
enum ApplyPart { AP_H = 1, AP_C = 2, AP_CT = 4, AP_G = 8, AP_GT = 16, AP_REG = 32, AP_ALL = 63 };

// Input / output vectors of an apply launch by vector space: x-space = [stagewise x | theta],
// y-space, z-space; a space no selected block reads (writes) may be null.  Strides are per problem,
// in doubles: one KKT vector [x | theta | y | z] has all three equal to its length.
struct ApplyIO {
  const double *x_x, *x_y, *x_z;
  double *y_x, *y_y, *y_z;
  long sx, sy, sz;
  int parts;
};

// y += K x (or the selected blocks of it), gather form: the lanes of a node item own the node's
// state rows of y_x, its dynamics rows and node constraint rows of y_y, and its rows of y_z; the
// lanes of an edge item own the control rows and the edge constraint rows.
__global__ void __launch_bounds__(TPB)
apply_kernel(const Meta mt, const double *__restrict__ model_all, const double *__restrict__ w_all,
             const double *__restrict__ r1_all, const double *__restrict__ r2_all,
             const double *__restrict__ r3_all, const ApplyIO io, long batch) {
  const int items = mt.N + mt.E;
  const long p = blockIdx.x / items;
  const int item = blockIdx.x - (unsigned)(p * items);
  if (p >= batch)
    return;
  const int parts = io.parts;
  const bool pH = parts & AP_H, pC = parts & AP_C, pCT = parts & AP_CT, pG = parts & AP_G, pGT = parts & AP_GT,
             pR = parts & AP_REG;
  const bool out_x = pH || pCT || pGT || pR, out_y = pC || pR, out_z = pG || pR;
  const int xt = mt.x_dim + mt.theta_dim; // theta (if any) sits behind the stagewise x
  const double *model = model_all + p * mt.model_len;
  // regularization diagonal (AP_REG only; null otherwise)
  const double *w = pR ? w_all + p * mt.z_dim : nullptr, *r1 = pR ? r1_all + p * xt : nullptr;
  const double *r2 = pR ? r2_all + p * mt.y_dim : nullptr, *r3 = pR ? r3_all + p * mt.z_dim : nullptr;
  const double *x_x = io.x_x ? io.x_x + p * io.sx : nullptr, *x_y = io.x_y ? io.x_y + p * io.sy : nullptr;
  const double *x_z = io.x_z ? io.x_z + p * io.sz : nullptr;
  double *y_x = io.y_x ? io.y_x + p * io.sx : nullptr, *y_y = io.y_y ? io.y_y + p * io.sy : nullptr;
  double *y_z = io.y_z ? io.y_z + p * io.sz : nullptr;
  const int tid = threadIdx.x;
  if (item < mt.N) {
    const int i = item, n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];
    const double *xs = x_x ? x_x + mt.x_state[i] : nullptr;
    const int lo = mt.child_offsets[i], hi = mt.child_offsets[i + 1];
    const int ine = mt.in_edge[i]; // -1 at the root
    for (int r = tid; r < 2 * n + c + g; r += TPB) {
      if (r < n) { // state rows of y_x: H x + C^T y + G^T z + r1 x
        if (!out_x)
          continue;
        const int d = r;
        double acc = 0.0;
        if (pH)
          acc = row_dot(model + mt.mo[N_Q][i], d, n, n, xs);
        if (pCT)
          acc += col_dot(model + mt.mo[N_JC][i], d, c, x_y + mt.y_node_c[i]);
        if (pGT)
          acc += col_dot(model + mt.mo[N_JG][i], d, g, x_z + mt.z_node[i]);
        if (pCT)
          acc -= x_y[mt.y_dyn[i] + d]; // -I of the node's own dynamics / initial-state row
        for (int ci = lo; ci < hi; ++ci) {
          const int e = mt.child_edges[ci], ch = mt.child[e], m = mt.cd[e];
          if (pH) {
            acc += row_dot(model + mt.mo[E_Q][e], d, n, n, xs);
            acc += row_dot(model + mt.mo[E_M][e], d, n, m, x_x + mt.x_control[e]);
          }
          if (pCT) {
            acc += col_dot(model + mt.mo[E_A][e], d, mt.sd[ch], x_y + mt.y_dyn[ch]);
            acc += col_dot(model + mt.mo[E_JXC][e], d, mt.ecd[e], x_y + mt.y_edge_c[e]);
          }
          if (pGT)
            acc += col_dot(model + mt.mo[E_JXG][e], d, mt.egd[e], x_z + mt.z_edge[e]);
        }
        const int at = mt.x_state[i] + d;
        y_x[at] += pR ? acc + r1[at] * x_x[at] : acc;
      } else if (r < 2 * n) { // dynamics rows of y_y
        if (!out_y)
          continue;
        const int d = r - n;
        double acc = 0.0;
        if (pC) {
          acc = -xs[d];
          if (ine >= 0) {
            const int e = ine, pa = mt.parent[e];
            acc += row_dot(model + mt.mo[E_A][e], d, n, mt.sd[pa], x_x + mt.x_state[pa]);
            acc += row_dot(model + mt.mo[E_B][e], d, n, mt.cd[e], x_x + mt.x_control[e]);
          }
        }
        const int at = mt.y_dyn[i] + d;
        y_y[at] += pR ? acc - r2[at] * x_y[at] : acc;
      } else if (r < 2 * n + c) {
        if (!out_y)
          continue;
        const int k = r - 2 * n, at = mt.y_node_c[i] + k;
        const double acc = pC ? row_dot(model + mt.mo[N_JC][i], k, c, n, xs) : 0.0;
        y_y[at] += pR ? acc - r2[at] * x_y[at] : acc;
      } else {
        if (!out_z)
          continue;
        const int k = r - 2 * n - c, at = mt.z_node[i] + k;
        const double acc = pG ? row_dot(model + mt.mo[N_JG][i], k, g, n, xs) : 0.0;
        y_z[at] += pR ? acc - (w[at] + r3[at]) * x_z[at] : acc;
      }
    }
  } else {
    const int e = item - mt.N, pa = mt.parent[e], ch = mt.child[e];
    const int n = mt.sd[pa], nc = mt.sd[ch], m = mt.cd[e], c = mt.ecd[e], g = mt.egd[e];
    const double *xp = x_x ? x_x + mt.x_state[pa] : nullptr, *ue = x_x ? x_x + mt.x_control[e] : nullptr;
    for (int r = tid; r < m + c + g; r += TPB) {
      if (r < m) { // control rows of y_x
        if (!out_x)
          continue;
        const int d = r;
        double acc = 0.0;
        if (pH) {
          acc = col_dot(model + mt.mo[E_M][e], d, n, xp);
          acc += row_dot(model + mt.mo[E_R][e], d, m, m, ue);
        }
        if (pCT) {
          acc += col_dot(model + mt.mo[E_B][e], d, nc, x_y + mt.y_dyn[ch]);
          acc += col_dot(model + mt.mo[E_JUC][e], d, c, x_y + mt.y_edge_c[e]);
        }
        if (pGT)
          acc += col_dot(model + mt.mo[E_JUG][e], d, g, x_z + mt.z_edge[e]);
        const int at = mt.x_control[e] + d;
        y_x[at] += pR ? acc + r1[at] * x_x[at] : acc;
      } else if (r < m + c) {
        if (!out_y)
          continue;
        const int k = r - m, at = mt.y_edge_c[e] + k;
        const double acc =
            pC ? row_dot(model + mt.mo[E_JXC][e], k, c, n, xp) + row_dot(model + mt.mo[E_JUC][e], k, c, m, ue) : 0.0;
        y_y[at] += pR ? acc - r2[at] * x_y[at] : acc;
      } else {
        if (!out_z)
          continue;
        const int k = r - m - c, at = mt.z_edge[e] + k;
        const double acc =
            pG ? row_dot(model + mt.mo[E_JXG][e], k, g, n, xp) + row_dot(model + mt.mo[E_JUG][e], k, g, m, ue) : 0.0;
        y_z[at] += pR ? acc - (w[at] + r3[at]) * x_z[at] : acc;
      }
    }
  }
}

} // namespace kkt
} // namespace sipamd
