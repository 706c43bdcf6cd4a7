// kkt_theta_chain_kernels.hpp -- the theta Schur complement of a UNIFORM CHAIN in fused passes
// (helpers.cpp:190-240 form_theta_jacobian, 372-407 the Schur complement, 414-747 the columns of K^-1 J_theta,
// 896-951 the solve).
//
// The generic path (kkt_theta_kernels.hpp) assembles J_theta column by column in memory, and every later pass
// (right-hand sides of the columns, multipliers of the columns, J^T K^-1 J, J^T K^-1 b) re-reads it.  J_theta is
// only a re-ordering of the model's theta blocks, and for a uniform chain the blocks of stage i (node i, edge i)
// are one contiguous item of the theta arena; so here the stage workgroups of the chain kernels stage that item
// in LDS beside the stage's constraint Jacobians and read J_theta's entries from it:
//
//   theta_rhs_chain_kernel      q_mod | c_mod | r_mod of all p columns        (was: theta_jacobian + condense<rhs>)
//   theta_recover_chain_kernel  x, u, y and the multipliers of all p columns -> K^-1 J_theta, AND the stage's
//                               share of  sum d2L_dtheta2 - J^T K^-1 J  (a p x p partial per stage)
//   theta_schur_reduce_kernel   sums the partials of a problem in stage order, adds r1_theta, LLT (:389-407)
//   theta_dot_chain_kernel      the stage's share of J^T (K^-1 b)              (solve, :920-928)
//
// J_theta itself is never written, K^-1 J_theta is written once and read only by the solve.
//
// Ownership of rows: stage i owns the state / control rows of node i and edge i, the constraint rows of node i
// and edge i, and the DYNAMICS row block of node i + 1 (J entries: ddyn_dtheta of edge i, in the stage's own
// item; the root's dynamics rows of J_theta are zero, :198).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kkt_chain_kernels.hpp"
#include "kkt_theta_kernels.hpp"

// wavefronts per SIMD the rhs / recover stage kernels are held to in the family instantiation (1: no cap; held to 8 --
// 64 registers -- `factor_theta` takes 2.67 ms against 2.54)
#ifndef SIP_KKT_THETA_WAVES
#define SIP_KKT_THETA_WAVES 1
#endif

namespace sipamd {
namespace kkt {

// theta arena of a uniform chain (sip_kkt_plan_set_theta lays the blocks of node i, then edge i, out in the order
// of ThetaBlock: N_X, N_C, N_G, N_TT | E_X, E_U, E_DYN, E_C, E_G, E_TT; every block rows x p, column-major)
struct ChainTheta {
  int p;                   // theta_dim
  long theta_len;          // doubles per problem
  int node_len, edge_len;  // (n + cn + gn + p) p of an interior node, (2 n + m + ce + ge + p) p of an edge
  int lds_item;            // doubles (even) of the largest stage item
};

struct ThetaStage { // the blocks of one stage's item (LDS)
  const double *NX, *NC, *NG, *NTT, *EX, *EU, *ED, *EC, *EG, *ETT;
  int n, m, c, g, ce, ge, p;
  bool last;
  // entry (row j of [x_i | u_i], column col) of J_theta: dL_dx_dtheta of node and edge (:222-223), dL_du_dtheta
  __device__ __forceinline__ double x(const int col, const int j) const {
    if (j < n)
      return last ? NX[j + n * col] : NX[j + n * col] + EX[j + n * col];
    return EU[(j - n) + m * col];
  }
  // ... of the dynamics rows of node i + 1
  __device__ __forceinline__ double dyn(const int col, const int j) const { return ED[j + n * col]; }
  // ... of constraint row k of the stage, rows ordered [node c | node g | edge c | edge g]
  __device__ __forceinline__ double row(const int col, const int k) const {
    if (k < c)
      return NC[k + c * col];
    if (k < c + g)
      return NG[(k - c) + g * col];
    if (k < c + g + ce)
      return EC[(k - c - g) + ce * col];
    return EG[(k - c - g - ce) + ge * col];
  }
  // sum over the rows the stage owns of J_theta[r, a] * v[r]; v given as [x_i | u_i], the dynamics rows of
  // node i + 1, the constraint rows
  __device__ __forceinline__ double dot(const int a, const double *vx, const double *vdyn, const double *vrow) const {
    double acc = 0.0;
    const int per = last ? n : n + m, nrows = c + g + ce + ge;
    for (int j = 0; j < per; ++j)
      acc += x(a, j) * vx[j];
    if (!last)
      for (int j = 0; j < n; ++j)
        acc += dyn(a, j) * vdyn[j];
    for (int k = 0; k < nrows; ++k)
      acc += row(a, k) * vrow[k];
    return acc;
  }
};

__device__ __forceinline__ ThetaStage theta_stage(const ChainKkt &ck, const ChainTheta &ct, const double *th,
                                                  const bool last) {
  ThetaStage s;
  const int n = ck.n, m = ck.m, p = ct.p;
  s.n = n, s.m = m, s.p = p, s.last = last;
  s.c = last ? ck.cT : ck.cn, s.g = last ? ck.gT : ck.gn;
  s.ce = last ? 0 : ck.ce, s.ge = last ? 0 : ck.ge;
  s.NX = th, s.NC = s.NX + n * p, s.NG = s.NC + s.c * p, s.NTT = s.NG + s.g * p;
  s.EX = s.NTT + p * p, s.EU = s.EX + n * p, s.ED = s.EU + m * p, s.EC = s.ED + n * p, s.EG = s.EC + s.ce * p,
  s.ETT = s.EG + s.ge * p;
  return s;
}

// doubles of stage i's item
__device__ __forceinline__ int theta_item_len(const ChainKkt &ck, const ChainTheta &ct, const bool last) {
  return last ? (ck.n + ck.cT + ck.gT + ct.p) * ct.p : ct.node_len + ct.edge_len;
}

// Three HBM -> LDS copies with the loads of all of them in flight before the first LDS store (as
// stage_copy_pair).  Falls back to consecutive copies when a range is unaligned or too long for one pass.
__device__ __forceinline__ void stage_copy_three(double *d0, const double *__restrict__ s0, int l0, double *d1,
                                                 const double *__restrict__ s1, int l1, double *d2,
                                                 const double *__restrict__ s2, int l2, int tid) {
  constexpr int U = 4, U2 = 8;
  auto vec_ok = [](const double *s, const double *d, int l, int u) { // (an empty range is fine: interior nodes
    return l == 0 ||                                                 // of the benchmark family have no constraints)
           (((l | (int)((uintptr_t)s >> 3) | (int)((uintptr_t)d >> 3)) & 1) == 0 && (l >> 1) <= u * TPB);
  };
  if (vec_ok(s0, d0, l0, U) && vec_ok(s1, d1, l1, U) && vec_ok(s2, d2, l2, U2)) {
    const d2_t *a2 = (const d2_t *)s0, *b2 = (const d2_t *)s1, *c2 = (const d2_t *)s2;
    d2_t *da = (d2_t *)d0, *db = (d2_t *)d1, *dc = (d2_t *)d2;
    const int n0 = l0 >> 1, n1 = l1 >> 1, n2 = l2 >> 1;
    d2_t va[U], vb[U], vc[U2];
    if (n0 > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        va[u] = a2[min(tid + u * TPB, n0 - 1)];
    }
    if (n1 > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        vb[u] = b2[min(tid + u * TPB, n1 - 1)];
    }
    if (n2 > 0) {
#pragma unroll
      for (int u = 0; u < U2; ++u)
        vc[u] = c2[min(tid + u * TPB, n2 - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (tid + u * TPB < n0)
        da[tid + u * TPB] = va[u];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (tid + u * TPB < n1)
        db[tid + u * TPB] = vb[u];
#pragma unroll
    for (int u = 0; u < U2; ++u)
      if (tid + u * TPB < n2)
        dc[tid + u * TPB] = vc[u];
  } else {
    stage_copy2(d0, s0, l0, tid);
    stage_copy2(d1, s1, l1, tid);
    stage_copy2(d2, s2, l2, tid);
  }
}

// (A software-pipelined walk of these stage kernels -- several consecutive items per wavefront, the next item's
// loads in flight in registers during the current item's sums, as condense_chain_pipe_kernel does -- was measured
// and is not here: at (12, 4), p = 8 it took 485 / 926 us for the rhs / recover kernel against 447 / 622 us for one
// item per workgroup; the prefetch registers (178 VGPRs) cost more occupancy than the overlap returns.)

// (Nor are they issue-bound: with theta_dim a template parameter as well (p = 4, 8 of the reference's theta benchmarks)
// and the scalar fallback of the stage copies compiled out, the static instruction count of the rhs / recover kernels
// halved -- 2 354 -> 1 167 and 2 298 -> 1 174 -- and `factor_theta` did not move (2.45 ms): they run at 4.2 / 4.9 TB/s
// of mostly 16-byte-per-lane traffic.  Not kept.)

// what every stage kernel below starts with
struct ThetaStageCtx {
  long p;
  int i, tid;
  bool last;
  int c, g, ce, ge, nrows;
  const double *item, *titem, *yinv, *zinv;
  int y_dyn, y_nc, y_ec, z_n, z_e;
};
__device__ __forceinline__ ThetaStageCtx theta_stage_ctx(const ChainKkt &ck, const ChainTheta &ct,
                                                         const double *model_all, const double *theta_all,
                                                         const double *inv_all) {
  ThetaStageCtx s;
  const int n = ck.n, T = ck.T;
  s.p = blockIdx.x / (T + 1);
  s.i = blockIdx.x - (unsigned)(s.p * (T + 1));
  s.tid = threadIdx.x;
  s.last = s.i == T;
  s.c = s.last ? ck.cT : ck.cn, s.g = s.last ? ck.gT : ck.gn;
  s.ce = s.last ? 0 : ck.ce, s.ge = s.last ? 0 : ck.ge;
  s.nrows = s.c + s.g + s.ce + s.ge;
  s.item = model_all != nullptr ? model_all + s.p * ck.model_len + (long)s.i * (ck.node_len + ck.edge_len) : nullptr;
  s.titem = theta_all + s.p * ct.theta_len + (long)s.i * (ct.node_len + ct.edge_len);
  s.yinv = inv_all != nullptr ? inv_all + s.p * ((long)ck.y_dim + ck.z_dim) : nullptr;
  s.zinv = s.yinv != nullptr ? s.yinv + ck.y_dim : nullptr;
  s.y_dyn = s.i * (n + ck.cn), s.y_nc = s.y_dyn + n, s.y_ec = T * (n + ck.cn) + n + ck.cT + s.i * ck.ce;
  s.z_n = s.i * ck.gn, s.z_e = T * ck.gn + ck.gT + s.i * ck.ge;
  return s;
}
// weight 1/r2, 1/(w + r3) of constraint row k of the stage
__device__ __forceinline__ double theta_weight_row(const ThetaStageCtx &s, const int k) {
  if (k < s.c)
    return s.yinv[s.y_nc + k];
  if (k < s.c + s.g)
    return s.zinv[s.z_n + (k - s.c)];
  if (k < s.c + s.g + s.ce)
    return s.yinv[s.y_ec + (k - s.c - s.g)];
  return s.zinv[s.z_e + (k - s.c - s.g - s.ce)];
}
// stagewise-KKT index of constraint row k (y rows from x_dim, z rows from x_dim + y_dim)
__device__ __forceinline__ long theta_row_at(const ChainKkt &ck, const ThetaStageCtx &s, const int k) {
  if (k < s.c)
    return (long)ck.x_dim + s.y_nc + k;
  if (k < s.c + s.g)
    return (long)ck.x_dim + ck.y_dim + s.z_n + (k - s.c);
  if (k < s.c + s.g + s.ce)
    return (long)ck.x_dim + s.y_ec + (k - s.c - s.g);
  return (long)ck.x_dim + ck.y_dim + s.z_e + (k - s.c - s.g - s.ce);
}

// q_mod | c_mod | r_mod (helpers.cpp:752-812) of ALL p columns of J_theta for stage i, the same sums in the same
// order as condense_chain_kernel<rhs only> forms them from an assembled J_theta.
// LDS: [Jacobian tails (lds_tail) | theta item (ct.lds_item) | weights (lds_rows) | weighted rows, p x lds_rows]
template <int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FN > 0 ? SIP_KKT_THETA_WAVES : 1, 8)))
theta_rhs_chain_kernel(const ChainKkt ck_in, const ChainTheta ct, const double *__restrict__ model_all,
                       const double *__restrict__ theta_all, const double *__restrict__ inv_all,
                       double *__restrict__ vecs_cols, const long vecs_col_stride,
                       const int32_t *__restrict__ status, const long batch) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  const ThetaStageCtx s = theta_stage_ctx(ck, ct, model_all, theta_all, inv_all);
  if (s.p >= batch || status[s.p] != 0)
    return;
  const int n = ck.n, m = ck.m, nn = n * n, nm = n * m, tid = s.tid, P = ct.p, R = ck.lds_rows;
  const int c = s.c, g = s.g, ce = s.ce, ge = s.ge, nrows = s.nrows;
  const bool last = s.last;
  double *jn = sm, *je = jn + (c + g) * n, *th = sm + ck.lds_tail, *wl = th + ct.lds_item, *wr = wl + R;
  const double w = tid < nrows ? theta_weight_row(s, tid) : 0.0;
  if (last) {
    stage_copy_three(jn, s.item + nn, (c + g) * n, je, s.item, 0, th, s.titem, theta_item_len(ck, ct, true), tid);
  } else {
    stage_copy_three(jn, s.item + nn, (c + g) * n, je, s.item + nn + (c + g) * n + 2 * nn + 2 * nm + m * m,
                     (ce + ge) * (n + m), th, s.titem, theta_item_len(ck, ct, false), tid);
  }
  if (tid < nrows)
    wl[tid] = w;
  for (int k = tid + TPB; k < nrows; k += TPB)
    wl[k] = theta_weight_row(s, k);
  __syncthreads();
  const ThetaStage ts = theta_stage(ck, ct, th, last);
  for (int e = tid; e < P * nrows; e += TPB) { // weights(constraint) * rhs(constraint), helpers.cpp:143
    const int col = e / nrows, k = e - col * nrows;
    wr[col * R + k] = wl[k] * ts.row(col, k);
  }
  __syncthreads();
  const double *Jc = jn, *Jg = jn + c * n;
  const double *Jxc = je, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  const int per = n + m;
  for (int e = tid; e < P * per; e += TPB) {
    const int col = e / per, j = e - col * per;
    if (last && j >= n)
      continue;
    double *vc = vecs_cols + col * vecs_col_stride + s.p * ck.vecs_len + (long)s.i * ck.vecs_stage;
    const double *wr_n = wr + col * R, *wr_e = wr_n + c + g;
    double acc = -ts.x(col, j);
    if (j < n) {
      acc = dot_seq<true>(acc, Jc + c * j, wr_n, c);
      acc = dot_seq<true>(acc, Jg + g * j, wr_n + c, g);
      if (!last) {
        acc = dot_seq<true>(acc, Jxc + ce * j, wr_e, ce);
        acc = dot_seq<true>(acc, Jxg + ge * j, wr_e + ce, ge);
      }
      vc[j] = acc;
      if (s.i == 0)
        vc[n + j] = -0.0; // the root's dynamics rows of J_theta are zero (:198)
      if (!last)
        vc[ck.vecs_stage + n + j] = -ts.dyn(col, j); // c_mod of node i + 1
    } else {
      const int d = j - n;
      acc = dot_seq<true>(acc, Juc + ce * d, wr_e, ce);
      acc = dot_seq<true>(acc, Jug + ge * d, wr_e + ce, ge);
      vc[2 * n + d] = acc;
    }
  }
}

// x, u, y scatter + multipliers (helpers.cpp:817-892) of ALL p columns for stage i -> K^-1 J_theta, and the stage's
// p x p share of  sum d2L_dtheta2 - J_theta^T K^-1 J_theta  (:389-398) -> s_part[problem][stage][a + p b].
// LDS: [Jacobian tails (lds_tail) | theta item | x_i|u_i of the columns, p (n + m) | y_{i+1} of the columns, p n |
//       multipliers of the columns, p x lds_rows | weights, lds_rows]
template <int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FN > 0 ? SIP_KKT_THETA_WAVES : 1, 8)))
theta_recover_chain_kernel(const ChainKkt ck_in, const ChainTheta ct, const double *__restrict__ model_all,
                           const double *__restrict__ theta_all, const double *__restrict__ inv_all,
                           const double *__restrict__ lqr_sol_cols, const long lqr_col_stride,
                           double *__restrict__ kj_cols, const long kj_col_stride, double *__restrict__ s_part,
                           const int32_t *__restrict__ status, const long batch) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  const ThetaStageCtx s = theta_stage_ctx(ck, ct, model_all, theta_all, inv_all);
  if (s.p >= batch || status[s.p] != 0)
    return;
  const int n = ck.n, m = ck.m, nn = n * n, nm = n * m, tid = s.tid, P = ct.p, R = ck.lds_rows;
  const int c = s.c, g = s.g, ce = s.ce, ge = s.ge, nrows = s.nrows, per = n + m;
  const bool last = s.last;
  const long kkt = (long)ck.x_dim + ck.y_dim + ck.z_dim;
  double *jn = sm, *je = jn + (c + g) * n, *th = sm + ck.lds_tail, *xs = th + ct.lds_item, *ys = xs + P * per,
         *ms = ys + P * n, *wl = ms + P * R;
  const double w = tid < nrows ? theta_weight_row(s, tid) : 0.0;
  // the stagewise solutions of the columns: requested before the stage copies (one HBM round trip, not two)
  constexpr int UX = 2;
  double v0[UX], v1[UX], v2[UX];
#pragma unroll
  for (int u = 0; u < UX; ++u) {
    const int e = tid + u * TPB, col = e / per, j = e - col * per;
    v0[u] = v1[u] = v2[u] = 0.0;
    if (e < P * per && !(last && j >= n)) {
      const double *ls = lqr_sol_cols + col * lqr_col_stride + s.p * ck.vecs_len + (long)s.i * ck.vecs_stage; // x | y | u
      v0[u] = j < n ? ls[j] : ls[2 * n + (j - n)];
      if (j < n) {
        v1[u] = ls[n + j];
        if (!last)
          v2[u] = ls[ck.vecs_stage + n + j]; // y of node i + 1
      }
    }
  }
  if (last) {
    stage_copy_three(jn, s.item + nn, (c + g) * n, je, s.item, 0, th, s.titem, theta_item_len(ck, ct, true), tid);
  } else {
    stage_copy_three(jn, s.item + nn, (c + g) * n, je, s.item + nn + (c + g) * n + 2 * nn + 2 * nm + m * m,
                     (ce + ge) * (n + m), th, s.titem, theta_item_len(ck, ct, false), tid);
  }
  auto put_x = [&](const int e, const double a0, const double a1, const double a2) {
    const int col = e / per, j = e - col * per;
    if (e >= P * per || (last && j >= n))
      return;
    double *sol = kj_cols + col * kj_col_stride + s.p * kkt, *sol_y = sol + ck.x_dim;
    xs[col * per + j] = a0;
    sol[s.i * (n + m) + j] = a0;
    if (j < n) {
      sol_y[s.y_dyn + j] = a1;
      ys[col * n + j] = a2;
    }
  };
  if (tid < nrows)
    wl[tid] = w;
  for (int k = tid + TPB; k < nrows; k += TPB)
    wl[k] = theta_weight_row(s, k);
#pragma unroll
  for (int u = 0; u < UX; ++u)
    put_x(tid + u * TPB, v0[u], v1[u], v2[u]);
  for (int e = tid + UX * TPB; e < P * per; e += TPB) { // more than 2 x 64 (column, row) pairs
    const int col = e / per, j = e - col * per;
    if (last && j >= n)
      continue;
    const double *ls = lqr_sol_cols + col * lqr_col_stride + s.p * ck.vecs_len + (long)s.i * ck.vecs_stage;
    put_x(e, j < n ? ls[j] : ls[2 * n + (j - n)], j < n ? ls[n + j] : 0.0,
          j < n && !last ? ls[ck.vecs_stage + n + j] : 0.0);
  }
  __syncthreads();
  const ThetaStage ts = theta_stage(ck, ct, th, last);
  const double *Jxc = je, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  for (int e = tid; e < P * nrows; e += TPB) { // the multipliers: (J x - rhs) * weight, one (column, row) pair per lane
    const int col = e / nrows, k = e - col * nrows;
    const double *xc = xs + col * per, *uc = xc + n;
    double jx;
    if (k < c)
      jx = row_dot(jn, k, c, n, xc);
    else if (k < c + g)
      jx = row_dot(jn + c * n, k - c, g, n, xc);
    else if (k < c + g + ce)
      jx = row_dot(Jxc, k - c - g, ce, n, xc) + row_dot(Juc, k - c - g, ce, m, uc);
    else
      jx = row_dot(Jxg, k - c - g - ce, ge, n, xc) + row_dot(Jug, k - c - g - ce, ge, m, uc);
    const double mult = (jx - ts.row(col, k)) * wl[k];
    ms[col * R + k] = mult;
    kj_cols[col * kj_col_stride + s.p * kkt + theta_row_at(ck, s, k)] = mult;
  }
  __syncthreads();
  const int pp = P * P;
  double *out = s_part + (s.p * (ck.T + 1) + s.i) * (long)pp;
  for (int q = tid; q < pp; q += TPB) { // entry (a, b): sum d2L_dtheta2 - J[:, a]^T (K^-1 J)[:, b] over the stage's rows
    const int b = q / P, a = q - b * P;
    const double base = last ? ts.NTT[q] : ts.NTT[q] + ts.ETT[q];
    out[q] = base - ts.dot(a, xs + b * per, ys + b * n, ms + b * R);
  }
}

// S = sum of the stage partials (in stage order) + diag(r1_theta), then LLT in place (helpers.cpp:399-407).
// One workgroup per problem.  LDS: p x p.
__global__ void __launch_bounds__(TPB)
theta_schur_reduce_kernel(const int nstages, const int p, const int sx, const double *__restrict__ r1_all,
                          const double *__restrict__ s_part, double *__restrict__ S_all,
                          int32_t *__restrict__ status, const long batch, const int fail_code) {
  extern __shared__ double sm[];
  const long prob = blockIdx.x;
  if (prob >= batch || status[prob] != 0)
    return;
  const int tid = threadIdx.x, pp = p * p;
  const double *part = s_part + prob * nstages * (long)pp;
  for (int q = tid; q < pp; q += TPB) {
    const int b = q / p, a = q - b * p;
    double acc[4] = {0.0, 0.0, 0.0, 0.0}; // four independent chains of loads; the sum order is fixed
    int i = 0;
    for (; i + 4 <= nstages; i += 4) {
      const double v0 = part[(long)i * pp + q], v1 = part[(long)(i + 1) * pp + q], v2 = part[(long)(i + 2) * pp + q],
                   v3 = part[(long)(i + 3) * pp + q];
      acc[0] += v0, acc[1] += v1, acc[2] += v2, acc[3] += v3;
    }
    for (; i < nstages; ++i)
      acc[0] += part[(long)i * pp + q];
    double v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (a == b)
      v += r1_all[prob * (sx + p) + sx + a];
    sm[q] = v;
  }
  __syncthreads();
  if (tid == 0) { // Eigen::LLT, lower, in place: pivot <= 0 -> NumericalIssue
    bool ok = true;
    for (int k = 0; k < p && ok; ++k) {
      double d = sm[k + p * k];
      for (int j = 0; j < k; ++j)
        d -= sm[k + p * j] * sm[k + p * j];
      if (!(d > 0.0)) {
        ok = false;
        break;
      }
      d = sqrt(d);
      sm[k + p * k] = d;
      for (int r = k + 1; r < p; ++r) {
        double v = sm[r + p * k];
        for (int j = 0; j < k; ++j)
          v -= sm[r + p * j] * sm[k + p * j];
        sm[r + p * k] = v / d;
      }
    }
    if (!ok)
      status[prob] = fail_code;
  }
  __syncthreads();
  for (int q = tid; q < pp; q += TPB)
    S_all[prob * pp + q] = sm[q];
}

// The stage's share of J_theta^T v for a stagewise vector v = [x | y | z] (the solve: v = K^-1 b, helpers.cpp:920-928)
// -> d_part[problem][stage][a].  LDS: [theta item | x_i|u_i rows of v (n + m) | dynamics rows of node i+1 (n) |
// constraint rows (lds_rows)]
template <int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB)
theta_dot_chain_kernel(const ChainKkt ck_in, const ChainTheta ct, const double *__restrict__ theta_all,
                       const double *__restrict__ v_all, double *__restrict__ d_part,
                       const int32_t *__restrict__ status, const long batch) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  const ThetaStageCtx s = theta_stage_ctx(ck, ct, nullptr, theta_all, nullptr);
  if (s.p >= batch || status[s.p] != 0)
    return;
  const int n = ck.n, m = ck.m, tid = s.tid, P = ct.p, per = n + m;
  const bool last = s.last;
  const long kkt = (long)ck.x_dim + ck.y_dim + ck.z_dim;
  double *th = sm, *vx = th + ct.lds_item, *vd = vx + per, *vr = vd + n;
  const double *v = v_all + s.p * kkt;
  // lanes 0 .. per-1: x_i | u_i; the next n: the dynamics rows of node i + 1; then the constraint rows
  double val = 0.0;
  int where = -1;
  {
    const int xl = last ? n : per, dl = last ? 0 : n;
    if (tid < xl)
      val = v[s.i * per + tid], where = tid;
    else if (tid - xl < dl)
      val = v[ck.x_dim + s.y_dyn + (n + ck.cn) + (tid - xl)], where = per + (tid - xl);
    else if (tid - xl - dl < s.nrows)
      val = v[theta_row_at(ck, s, tid - xl - dl)], where = per + n + (tid - xl - dl);
  }
  stage_copy2(th, s.titem, theta_item_len(ck, ct, last), tid);
  if (where >= 0)
    vx[where] = val;
  {
    const int used = (last ? n : per + n) + s.nrows; // more rows than lanes: the rest in a loop
    for (int e = tid + TPB; e < used; e += TPB) {
      const int xl = last ? n : per, dl = last ? 0 : n;
      if (e < xl)
        vx[e] = v[s.i * per + e];
      else if (e - xl < dl)
        vd[e - xl] = v[ck.x_dim + s.y_dyn + (n + ck.cn) + (e - xl)];
      else
        vr[e - xl - dl] = v[theta_row_at(ck, s, e - xl - dl)];
    }
  }
  __syncthreads();
  const ThetaStage ts = theta_stage(ck, ct, th, last);
  for (int a = tid; a < P; a += TPB)
    d_part[(s.p * (ck.T + 1) + s.i) * (long)P + a] = ts.dot(a, vx, vd, vr);
}

// The theta terms of y += K x, or of the selected blocks of it (ApplyIO::parts: the theta sections of add_Hx / Cx / CTx /
// Gx / GTx_to_y, helpers.cpp:1023-1066, 1128-1158, 1221-1249, 1285-1308, 1344-1367), for a uniform chain.  The generic
// kernel (apply_theta_kernel) walks the nodes and edges of a problem one after the other with one wavefront (2.4 ms at
// the benchmark shape, p = 8, for 0.94 GB of theta arena).  Here a workgroup of W wavefronts owns a problem, wavefront w
// takes stages w, w + W, ...: it stages the stage's theta item and the stage's slices of x in ITS part of the LDS, adds
// J_theta theta to the rows the stage owns, and accumulates the stage's share of y_theta = J_theta^T x + H_tt theta in
// its LDS accumulator; the W accumulators are summed in wavefront order at the end (no scratch memory, no atomics:
// the result does not depend on scheduling).
// LDS per wavefront: [theta item | x_i|u_i (n + m) | dynamics rows of node i+1 (n) | constraint rows (lds_rows) |
//                     theta (p, even) | y_theta accumulator (p, even)]
template <int FN = 0, int FM = 0>
__global__ void __launch_bounds__(512)
apply_theta_chain_kernel(const ChainKkt ck_in, const ChainTheta ct, const double *__restrict__ theta_all,
                         const double *__restrict__ r1_all, const ApplyIO io, const long batch) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  const long prob = blockIdx.x;
  if (prob >= batch)
    return;
  const int n = ck.n, m = ck.m, T = ck.T, P = ct.p, per = n + m, R = ck.lds_rows, Pe = (P + 1) & ~1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const int vlen = (per + n + 1) & ~1;                // x_i|u_i and the dynamics rows, padded to an even length
  const int wl = ct.lds_item + vlen + R + 2 * Pe;     // doubles per wavefront (even: the items stay 16-byte aligned)
  double *th = sm + wave * wl, *vx = th + ct.lds_item, *vd = vx + per, *vr = vx + vlen, *tv = vr + R, *yacc = tv + Pe;
  const int parts = io.parts;
  const bool pH = parts & AP_H, pC = parts & AP_C, pCT = parts & AP_CT, pG = parts & AP_G, pGT = parts & AP_GT,
             pR = parts & AP_REG;
  const int sx = ck.x_dim;
  const double *x_x = io.x_x ? io.x_x + prob * io.sx : nullptr, *x_y = io.x_y ? io.x_y + prob * io.sy : nullptr;
  const double *x_z = io.x_z ? io.x_z + prob * io.sz : nullptr;
  double *y_x = io.y_x ? io.y_x + prob * io.sx : nullptr, *y_y = io.y_y ? io.y_y + prob * io.sy : nullptr;
  double *y_z = io.y_z ? io.y_z + prob * io.sz : nullptr;
  const bool use_theta = (pH || pC || pG || pR) && x_x != nullptr; // theta is read by H, C, G (and the r1 term)
  for (int a = lane; a < P; a += 64) {
    tv[a] = use_theta ? x_x[sx + a] : 0.0;
    yacc[a] = 0.0;
  }
  for (int i = wave; i <= T; i += waves) {
    ThetaStageCtx s;
    {
      s.p = prob, s.i = i, s.tid = lane, s.last = i == T;
      s.c = s.last ? ck.cT : ck.cn, s.g = s.last ? ck.gT : ck.gn;
      s.ce = s.last ? 0 : ck.ce, s.ge = s.last ? 0 : ck.ge;
      s.nrows = s.c + s.g + s.ce + s.ge;
      s.item = nullptr, s.yinv = s.zinv = nullptr;
      s.titem = theta_all + prob * ct.theta_len + (long)i * (ct.node_len + ct.edge_len);
      s.y_dyn = i * (n + ck.cn), s.y_nc = s.y_dyn + n, s.y_ec = T * (n + ck.cn) + n + ck.cT + i * ck.ce;
      s.z_n = i * ck.gn, s.z_e = T * ck.gn + ck.gT + i * ck.ge;
    }
    const bool last = s.last;
    const int c = s.c, g = s.g, ce = s.ce, nrows = s.nrows, xl = last ? n : per;
    // constraint row k of the stage: in the y-space (c rows) or the z-space (g rows), and where
    auto row_is_y = [&](const int k) { return k < c || (k >= c + g && k < c + g + ce); };
    auto row_at = [&](const int k) {
      if (k < c)
        return s.y_nc + k;
      if (k < c + g)
        return s.z_n + (k - c);
      if (k < c + g + ce)
        return s.y_ec + (k - c - g);
      return s.z_e + (k - c - g - ce);
    };
    // Every small read of the stage -- the lane's entries of the slices of x, and the present value of the output
    // rows the lane will update -- is requested before the item's copy: one HBM round trip per stage, not one per
    // slice and another per read-modify-write.  (Entries past the first 64 of a slice, if any, take the plain path.)
    const bool yrow = lane < nrows && row_is_y(lane);
    const int rat = lane < nrows ? row_at(lane) : 0;
    double r_vx = 0.0, r_vd = 0.0, r_vr = 0.0, o_x = 0.0, o_d = 0.0, o_r = 0.0;
    if (lane < xl) {
      if (pH)
        r_vx = x_x[i * per + lane], o_x = y_x[i * per + lane];
    }
    if (!last && lane < n) {
      if (pCT)
        r_vd = x_y[s.y_dyn + (n + ck.cn) + lane];
      if (pC)
        o_d = y_y[s.y_dyn + (n + ck.cn) + lane];
    }
    if (lane < nrows) {
      if (yrow ? pCT : pGT)
        r_vr = (yrow ? x_y : x_z)[rat];
      if (yrow ? pC : pG)
        o_r = (yrow ? y_y : y_z)[rat];
    }
    stage_copy2(th, s.titem, theta_item_len(ck, ct, last), lane);
    // the stage's slices of x in LDS, zero where the selected blocks do not read them
    if (lane < xl)
      vx[lane] = r_vx;
    if (!last && lane < n)
      vd[lane] = r_vd;
    if (lane < nrows)
      vr[lane] = r_vr;
    for (int e = lane + 64; e < xl; e += 64)
      vx[e] = pH ? x_x[i * per + e] : 0.0;
    if (!last)
      for (int e = lane + 64; e < n; e += 64)
        vd[e] = pCT ? x_y[s.y_dyn + (n + ck.cn) + e] : 0.0;
    for (int k = lane + 64; k < nrows; k += 64) {
      const bool iny = row_is_y(k);
      vr[k] = iny ? (pCT ? x_y[row_at(k)] : 0.0) : (pGT ? x_z[row_at(k)] : 0.0);
    }
    const ThetaStage ts = theta_stage(ck, ct, th, last);
    // rows += J_theta theta
    if (pH) {
      for (int j = lane; j < xl; j += 64) {
        double *dst = y_x + i * per + j;
        const double old = j == lane ? o_x : *dst;
        if (j < n) { // the node's block, then the edge's (the order of the node / edge loops of the reference)
          double an = 0.0, ae = 0.0;
          for (int a = 0; a < P; ++a)
            an += ts.NX[j + n * a] * tv[a];
          if (!last)
            for (int a = 0; a < P; ++a)
              ae += ts.EX[j + n * a] * tv[a];
          *dst = last ? old + an : (old + an) + ae;
        } else {
          double au = 0.0;
          for (int a = 0; a < P; ++a)
            au += ts.EU[(j - n) + m * a] * tv[a];
          *dst = old + au;
        }
      }
    }
    if (pC && !last)
      for (int j = lane; j < n; j += 64) { // dynamics rows of node i + 1
        double acc = 0.0;
        for (int a = 0; a < P; ++a)
          acc += ts.ED[j + n * a] * tv[a];
        double *dst = y_y + s.y_dyn + (n + ck.cn) + j;
        *dst = (j == lane ? o_d : *dst) + acc;
      }
    if (pC || pG)
      for (int k = lane; k < nrows; k += 64) {
        const bool iny = row_is_y(k);
        if (iny ? !pC : !pG)
          continue;
        double acc = 0.0;
        for (int a = 0; a < P; ++a)
          acc += ts.row(a, k) * tv[a];
        double *dst = (iny ? y_y : y_z) + row_at(k);
        *dst = (k == lane ? o_r : *dst) + acc;
      }
    // y_theta += J_theta^T x (H, CT, GT) + H_theta_theta theta (H)
    if (pH || pCT || pGT)
      for (int a = lane; a < P; a += 64) {
        double acc = ts.dot(a, vx, vd, vr);
        if (pH)
          for (int b = 0; b < P; ++b)
            acc += (last ? ts.NTT[a + P * b] : ts.NTT[a + P * b] + ts.ETT[a + P * b]) * tv[b];
        yacc[a] += acc;
      }
  }
  __syncthreads();
  if (wave == 0 && (pH || pCT || pGT || pR) && y_x != nullptr)
    for (int a = lane; a < P; a += 64) {
      double acc = 0.0;
      for (int w = 0; w < waves; ++w)
        acc += (sm + w * wl + (wl - Pe))[a];
      if (pR)
        acc += r1_all[prob * (sx + P) + sx + a] * tv[a];
      y_x[sx + a] += acc;
    }
}

// theta = S^-1 (b_theta - J^T K^-1 b) with J^T K^-1 b from the stage partials; sol = K^-1 b - (K^-1 J) theta,
// re-inserted as [x | theta | y | z]  (helpers.cpp:920-950).  One workgroup per problem.  LDS: p.
__global__ void __launch_bounds__(TPB)
theta_finish_parts_kernel(const int nstages, const int p, const int sx, const long skkt,
                          const double *__restrict__ b_all, const double *__restrict__ d_part,
                          const double *__restrict__ KJ_all, const double *__restrict__ S_all,
                          const double *__restrict__ sw_all, double *__restrict__ sol_all,
                          const int32_t *__restrict__ status, const long batch) {
  extern __shared__ double sm[]; // theta (p)
  const long prob = blockIdx.x;
  if (prob >= batch || status[prob] != 0)
    return;
  const int tid = threadIdx.x;
  const long col_stride = batch * skkt;
  const double *KJp = KJ_all + prob * skkt, *sw = sw_all + prob * skkt;
  const double *b_theta = b_all + prob * (skkt + p) + sx;
  double *sol = sol_all + prob * (skkt + p);
  const double *part = d_part + prob * nstages * (long)p;
  for (int a = tid; a < p; a += TPB) {
    double dot = 0.0;
    for (int i = 0; i < nstages; ++i)
      dot += part[(long)i * p + a];
    sm[a] = b_theta[a] - dot;
  }
  __syncthreads();
  if (tid == 0) {
    const double *L = S_all + prob * p * p;
    for (int i = 0; i < p; ++i) {
      double v = sm[i];
      for (int j = 0; j < i; ++j)
        v -= L[i + p * j] * sm[j];
      sm[i] = v / L[i + p * i];
    }
    for (int i = p - 1; i >= 0; --i) {
      double v = sm[i];
      for (int j = i + 1; j < p; ++j)
        v -= L[j + p * i] * sm[j];
      sm[i] = v / L[i + p * i];
    }
  }
  __syncthreads();
  for (long r = tid; r < skkt; r += TPB) {
    double acc = 0.0;
    for (int a = 0; a < p; ++a)
      acc += KJp[a * col_stride + r] * sm[a];
    sol[r < sx ? r : r + p] = sw[r] - acc;
  }
  for (int a = tid; a < p; a += TPB)
    sol[sx + a] = sm[a];
}

} // namespace kkt
} // namespace sipamd
