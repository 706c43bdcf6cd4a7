// kkt_chain_kernels.hpp -- the Newton-KKT condensation / recovery kernels for
// UNIFORM CHAINS (the shape family of benchmarks/newton_kkt_benchmark.cpp:
// 58-83 and of stagewise optimal control in general): every node has n
// states, every edge m controls and (ce, ge) constraint rows, interior nodes
// (cn, gn) rows and the terminal node (cT, gT).
//
// One workgroup (one wavefront) per (problem, stage i): node i and edge i,
// which are adjacent in the model arena, travel HBM -> LDS as one coalesced
// copy; every address is computed from a handful of scalars instead of being
// fetched from offset tables; the rank-(c+g) symmetric updates J^T diag(w) J
// of Q_mod / M_mod / R_mod (helpers.cpp:79-136) run as 16 x 16 x 4 tiles on
// the fp64 matrix pipe.  The products are summed in the MFMA's order rather
// than constraint by constraint, so results agree with the table-driven
// kernels to rounding (~1e-15 relative), not bitwise.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kkt_kernels.hpp"

namespace sipamd {
namespace kkt {

struct ChainKkt {
  int n, m, T, cn, gn, cT, gT, ce, ge;
  int node_len, edge_len; // model doubles of an interior node item / an edge item
  long model_len;
  int x_dim, y_dim, z_dim;
  int mats_stage, vecs_stage; // packed chain layout strides (include/sip_lqr_amd.h)
  long mats_len, vecs_len;
  // LDS plan (doubles, even): one whole stage of the model (condense) | every
  // Jacobian of a stage (recover) | constraint rows of a stage
  int lds_item, lds_tail, lds_rows;
};

typedef double d2_t __attribute__((ext_vector_type(2)));

// acc (+/-)= sum_k a[k] * b[k], k ascending (the order of the direct kernels)
template <bool NEG>
__device__ __forceinline__ double dot_seq(double acc, const double *a, const double *b, int rows) {
  if (((rows | (int)((uintptr_t)a >> 3) | (int)((uintptr_t)b >> 3)) & 1) == 0) {
    const d2_t *a2 = (const d2_t *)a, *b2 = (const d2_t *)b;
    for (int k = 0; k < (rows >> 1); ++k) {
      const d2_t av = a2[k], bv = b2[k];
      if (NEG) {
        acc -= av[0] * bv[0];
        acc -= av[1] * bv[1];
      } else {
        acc += bv[0] * av[0];
        acc += bv[1] * av[1];
      }
    }
  } else {
    for (int k = 0; k < rows; ++k) {
      if (NEG)
        acc -= a[k] * b[k];
      else
        acc += b[k] * a[k];
    }
  }
  return acc;
}

// HBM -> LDS copy of `len` doubles by one wavefront.  All the loads of a pass
// are issued before the first LDS store (a plain copy loop waits for each
// load in turn: one HBM round trip per 1 KB).
__device__ __forceinline__ void stage_copy2(double *dst, const double *__restrict__ src, int len, int tid) {
  constexpr int U = 8;
  // 16-byte pieces when both ends allow it (len even, both 16-byte aligned)
  if (((len | (int)((uintptr_t)src >> 3) | (int)((uintptr_t)dst >> 3)) & 1) == 0) {
    const d2_t *s2 = (const d2_t *)src;
    d2_t *d2 = (d2_t *)dst;
    const int len2 = len >> 1;
    for (int base = tid; base < len2; base += U * TPB) {
      d2_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (base + u * TPB < len2)
          v[u] = s2[base + u * TPB];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (base + u * TPB < len2)
          d2[base + u * TPB] = v[u];
    }
  } else {
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
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// WITH_RHS: also q_mod / r_mod / c_mod from b.  MATS = false (the split solve path): ONLY those --
// the Jacobians of the stage are staged, no Q_mod / M_mod / R_mod / A / B work, `status` problems
// with a failed factorization are skipped.
template <bool WITH_RHS, bool MATS = true>
__global__ void __launch_bounds__(TPB)
condense_chain_kernel(const ChainKkt ck, const double *__restrict__ model_all, const double *__restrict__ r1_all,
                      const double *__restrict__ inv_all, double *__restrict__ mats_all,
                      const double *__restrict__ b_all, double *__restrict__ vecs_all, long batch,
                      const int32_t *__restrict__ status = nullptr, const int ncols = 1,
                      const long b_col_stride = 0, const long vecs_col_stride = 0) {
  // ncols > 1 (the columns of J_theta, helpers.cpp:414-747): column j's right-hand side is
  // b_all + j * b_col_stride, its q_mod / r_mod / c_mod go to vecs_all + j * vecs_col_stride; the
  // Jacobians of the stage are staged once for all columns.
  static_assert(MATS || WITH_RHS, "nothing to do");
  extern __shared__ double sm[];
  const int n = ck.n, m = ck.m, T = ck.T;
  double *buf = sm, *wl = buf + ck.lds_item, *wr = wl + ck.lds_rows;
  const long p = blockIdx.x / (T + 1);
  const int i = blockIdx.x - (unsigned)(p * (T + 1));
  if (p >= batch || (!MATS && status != nullptr && status[p] != 0))
    return;
  const int tid = threadIdx.x;
  const bool last = i == T;
  const int c = last ? ck.cT : ck.cn, g = last ? ck.gT : ck.gn;
  const int ce = last ? 0 : ck.ce, ge = last ? 0 : ck.ge;
  const int nn = n * n, nm = n * m;
  const int node_len = nn + (c + g) * n;
  const long kkt = (long)ck.x_dim + ck.y_dim + ck.z_dim;
  const double *item = model_all + p * ck.model_len + (long)i * (ck.node_len + ck.edge_len);
  const double *r1 = r1_all + p * ck.x_dim + i * (n + m);
  const double *yinv = inv_all + p * ((long)ck.y_dim + ck.z_dim), *zinv = yinv + ck.y_dim;
  // flattened orderings (types.cpp:24-64) of a uniform chain
  const int y_dyn = i * (n + ck.cn), y_nc = y_dyn + n, y_ec = T * (n + ck.cn) + n + ck.cT + i * ck.ce;
  const int z_n = i * ck.gn, z_e = T * ck.gn + ck.gT + i * ck.ge;
  const double *b = WITH_RHS ? b_all + p * kkt : nullptr;
  const double *b_y = WITH_RHS ? b + ck.x_dim : nullptr, *b_z = WITH_RHS ? b_y + ck.y_dim : nullptr;
  double *mats = mats_all + p * ck.mats_len + (long)i * ck.mats_stage;
  double *vecs = WITH_RHS ? vecs_all + p * ck.vecs_len + (long)i * ck.vecs_stage : nullptr;

  // whole stage (node item + edge item are adjacent in the model arena) -> LDS,
  // and the weights 1/r2, 1/(w+r3) of its constraint rows: [node c | node g | edge c | edge g]
  const int edge_len = last ? 0 : ck.edge_len;
  if (MATS) {
    stage_copy2(buf, item, node_len + edge_len, tid);
  } else { // the Jacobian tails only, at their usual places in the image
    stage_copy2(buf + nn, item + nn, (c + g) * n, tid);
    if (!last) {
      const int o_tail = node_len + 2 * nn + 2 * nm + m * m;
      stage_copy2(buf + o_tail, item + o_tail, (ce + ge) * (n + m), tid);
    }
  }
  const int nrows = c + g + ce + ge;
  // entry of the right-hand side that belongs to constraint row k of this stage
  auto rhs_row = [&](const int k, const double *by, const double *bz) {
    if (k < c)
      return by[y_nc + k];
    if (k < c + g)
      return bz[z_n + (k - c)];
    if (k < c + g + ce)
      return by[y_ec + (k - c - g)];
    return bz[z_e + (k - c - g - ce)];
  };
  for (int k = tid; k < nrows; k += TPB) {
    double wk;
    if (k < c)
      wk = yinv[y_nc + k];
    else if (k < c + g)
      wk = zinv[z_n + (k - c)];
    else if (k < c + g + ce)
      wk = yinv[y_ec + (k - c - g)];
    else
      wk = zinv[z_e + (k - c - g - ce)];
    wl[k] = wk;
    if (WITH_RHS)
      wr[k] = wk * rhs_row(k, b_y, b_z); // weights(constraint) * rhs(constraint), helpers.cpp:143
  }
  const double *Jc = buf + nn, *Jg = Jc + c * n;
  const double *eb = buf + node_len; // edge item
  const int o_m = nn, o_r = o_m + nm, o_a = o_r + m * m, o_b = o_a + nn, o_j = o_b + nm;
  const double *Jxc = eb + o_j, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  __syncthreads();

  if (MATS)
  // ---- Q_mod, M_mod, R_mod (helpers.cpp:299-361) ----
  // T = J^T diag(w) J over the stage's combined columns [x (n) | u (m)] and all its constraint
  // rows, as 16 x 16 tiles on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64: lane l feeds
  // A[l & 15][k = l >> 4] and B[k = l >> 4][l & 15], gets D[(l >> 4) + 4 r][l & 15]).  Its blocks:
  // T[x, x] -> Q_mod, T[x, u] -> M_mod, T[u, u] -> R_mod (lower tiles only; mirrored on the way out).
  {
    const int li = tid & 15, lg = tid >> 4;
    const int ncols = last ? n : n + m;
    const int nt = (ncols + 15) >> 4;
    double *Qm = mats, *Am = mats + nn + n, *Bm = Am + nn, *Mm = Bm + nm, *Rm = Mm + nm;
    for (int ti = 0; ti < nt; ++ti) {
      for (int tj = 0; tj <= ti; ++tj) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        const int ci = ti * 16 + li, cj = tj * 16 + li;
        auto row_block = [&](const double *J, const double *wgt, int rows, int cols) {
          for (int k0 = 0; k0 < rows; k0 += 4) {
            const int k = k0 + lg;
            const bool ok = k < rows;
            const double a = (ok && ci < cols) ? J[k + rows * ci] : 0.0;
            const double bj = (ok && cj < cols) ? wgt[k] * J[k + rows * cj] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bj, acc, 0, 0, 0);
          }
        };
        row_block(Jc, wl, c, n);
        row_block(Jg, wl + c, g, n);
        row_block(Jxc, wl + c + g, ce, n + m); // [dc_dx | dc_du]
        row_block(Jxg, wl + c + g + ce, ge, n + m);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int I = ti * 16 + lg + 4 * r, J = cj;
          const double t = acc[r];
          if (I < n) {
            if (J <= I) { // Q_mod, lower, mirrored (:310-318, 336-339, 357-361)
              double v = buf[I + n * J];
              if (I == J)
                v += r1[I];
              if (!last)
                v += eb[I + n * J];
              v += t;
              Qm[I + n * J] = v;
              Qm[J + n * I] = v;
            }
          } else if (I < ncols) {
            const int u = I - n;
            if (J < n) { // M_mod(x_J, u), :341-352
              Mm[J + n * u] = eb[o_m + J + n * u] + t;
            } else if (J - n <= u) { // R_mod, lower, mirrored (:344-354)
              const int uj = J - n;
              double v = eb[o_r + u + m * uj];
              if (u == uj)
                v += r1[n + u];
              v += t;
              Rm[u + m * uj] = v;
              Rm[uj + m * u] = v;
            }
          }
        }
      }
    }
    if (tid < n)
      mats[nn + tid] = yinv[y_dyn + tid]; // dyn_r2
    if (!last) {
      for (int k = tid; k < nn; k += TPB) // ddyn_dx, ddyn_du, :365-366
        Am[k] = eb[o_a + k];
      for (int k = tid; k < nm; k += TPB)
        Bm[k] = eb[o_b + k];
    }
  }
  if (WITH_RHS) { // q_mod, c_mod, r_mod (helpers.cpp:752-812)
    const double *wr_n = wr, *wr_e = wr + c + g;
    for (int col = 0; col < ncols; ++col) {
      const double *bc = b + col * b_col_stride, *bc_y = bc + ck.x_dim, *bc_z = bc_y + ck.y_dim;
      double *vc = vecs + col * vecs_col_stride;
      if (col > 0) { // weighted right-hand side rows of this column
        __syncthreads();
        for (int k = tid; k < nrows; k += TPB)
          wr[k] = wl[k] * rhs_row(k, bc_y, bc_z);
        __syncthreads();
      }
      if (tid < n) {
        double acc = -bc[i * (n + m) + tid];
        acc = dot_seq<true>(acc, Jc + c * tid, wr_n, c);
        acc = dot_seq<true>(acc, Jg + g * tid, wr_n + c, g);
        if (!last) {
          acc = dot_seq<true>(acc, Jxc + ce * tid, wr_e, ce);
          acc = dot_seq<true>(acc, Jxg + ge * tid, wr_e + ce, ge);
        }
        vc[tid] = acc;
        vc[n + tid] = -bc_y[y_dyn + tid];
      } else if (!last && tid >= 32 && tid - 32 < m) { // another half of the wave: r_mod
        const int d = tid - 32;
        double acc = -bc[i * (n + m) + n + d];
        acc = dot_seq<true>(acc, Juc + ce * d, wr_e, ce);
        acc = dot_seq<true>(acc, Jug + ge * d, wr_e + ce, ge);
        vc[2 * n + d] = acc;
      }
    }
  }
}

// x, u, y scatter + multipliers of node i and edge i (helpers.cpp:817-892).
__global__ void __launch_bounds__(TPB)
recover_chain_kernel(const ChainKkt ck, const double *__restrict__ model_all, const double *__restrict__ b_all,
                     const double *__restrict__ inv_all, const double *__restrict__ lqr_sol_all,
                     double *__restrict__ sol_all, const int32_t *__restrict__ status, long batch,
                     const int ncols = 1, const long b_col_stride = 0, const long lqr_col_stride = 0,
                     const long sol_col_stride = 0) {
  extern __shared__ double sm[];
  const int n = ck.n, m = ck.m, T = ck.T;
  const long p = blockIdx.x / (T + 1);
  const int i = blockIdx.x - (unsigned)(p * (T + 1));
  if (p >= batch || status[p] != 0)
    return;
  const int tid = threadIdx.x;
  const bool last = i == T;
  const int c = last ? ck.cT : ck.cn, g = last ? ck.gT : ck.gn;
  const int ce = last ? 0 : ck.ce, ge = last ? 0 : ck.ge;
  const int nn = n * n, nm = n * m;
  const long kkt = (long)ck.x_dim + ck.y_dim + ck.z_dim;
  const double *item = model_all + p * ck.model_len + (long)i * (ck.node_len + ck.edge_len);
  const double *yinv = inv_all + p * ((long)ck.y_dim + ck.z_dim), *zinv = yinv + ck.y_dim;
  const int y_dyn = i * (n + ck.cn), y_nc = y_dyn + n, y_ec = T * (n + ck.cn) + n + ck.cT + i * ck.ce;
  const int z_n = i * ck.gn, z_e = T * ck.gn + ck.gT + i * ck.ge;
  double *jn = sm, *je = jn + (c + g) * n, *xs = sm + ck.lds_tail, *us = xs + n;
  stage_copy2(jn, item + nn, (c + g) * n, tid); // [dc_dx | dg_dx] of the node
  if (!last)
    stage_copy2(je, item + nn + (c + g) * n + 2 * nn + 2 * nm + m * m, (ce + ge) * (n + m), tid);
  const double *Jxc = je, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  for (int col = 0; col < ncols; ++col) { // ncols > 1: the columns of K^-1 J_theta, Jacobians staged once
    const double *b_y = b_all + col * b_col_stride + p * kkt + ck.x_dim, *b_z = b_y + ck.y_dim;
    const double *ls = lqr_sol_all + col * lqr_col_stride + p * ck.vecs_len + (long)i * ck.vecs_stage; // x_i | y_i | u_i
    double *sol = sol_all + col * sol_col_stride + p * kkt, *sol_y = sol + ck.x_dim, *sol_z = sol_y + ck.y_dim;
    if (col > 0)
      __syncthreads(); // the previous column's readers of xs / us are done
    if (tid < n) {
      const double xv = ls[tid];
      xs[tid] = xv;
      sol[i * (n + m) + tid] = xv;
      sol_y[y_dyn + tid] = ls[n + tid];
    } else if (!last && tid >= 32 && tid - 32 < m) {
      const double uv = ls[2 * n + (tid - 32)];
      us[tid - 32] = uv;
      sol[i * (n + m) + n + (tid - 32)] = uv;
    }
    __syncthreads();
    for (int k = tid; k < c + g; k += TPB) {
      if (k < c)
        sol_y[y_nc + k] = (row_dot(jn, k, c, n, xs) - b_y[y_nc + k]) * yinv[y_nc + k];
      else
        sol_z[z_n + (k - c)] = (row_dot(jn + c * n, k - c, g, n, xs) - b_z[z_n + (k - c)]) * zinv[z_n + (k - c)];
    }
    for (int k = tid; k < ce + ge; k += TPB) {
      if (k < ce) {
        const double jx = row_dot(Jxc, k, ce, n, xs), ju = row_dot(Juc, k, ce, m, us);
        sol_y[y_ec + k] = ((jx + ju) - b_y[y_ec + k]) * yinv[y_ec + k];
      } else {
        const int kk = k - ce;
        const double jx = row_dot(Jxg, kk, ge, n, xs), ju = row_dot(Jug, kk, ge, m, us);
        sol_z[z_e + kk] = ((jx + ju) - b_z[z_e + kk]) * zinv[z_e + kk];
      }
    }
  }
}

// y += K x (helpers.cpp:953-1368) for a uniform chain.  The workgroup of stage i owns the state
// and control rows of stage i, the dynamics rows of node i + 1 (plus the root's at i = 0) and the
// constraint rows of node i and edge i: everything they touch is in stage i's model item (staged
// in LDS) and in a few slices of x.  Vectors are [x | theta (th entries) | y | z].
__global__ void __launch_bounds__(TPB)
apply_chain_kernel(const ChainKkt ck, const int th, const double *__restrict__ model_all,
                   const double *__restrict__ w_all, const double *__restrict__ r1_all,
                   const double *__restrict__ r2_all, const double *__restrict__ r3_all,
                   const double *__restrict__ x_all, double *__restrict__ y_all, long batch) {
  extern __shared__ double sm[];
  const int n = ck.n, m = ck.m, T = ck.T;
  const long p = blockIdx.x / (T + 1);
  const int i = blockIdx.x - (unsigned)(p * (T + 1));
  if (p >= batch)
    return;
  const int tid = threadIdx.x;
  const bool last = i == T;
  const int c = last ? ck.cT : ck.cn, g = last ? ck.gT : ck.gn;
  const int ce = last ? 0 : ck.ce, ge = last ? 0 : ck.ge;
  const int nn = n * n, nm = n * m;
  const int node_len = nn + (c + g) * n, edge_len = last ? 0 : ck.edge_len;
  const int xt = ck.x_dim + th;
  const long full = (long)xt + ck.y_dim + ck.z_dim;
  const double *item = model_all + p * ck.model_len + (long)i * (ck.node_len + ck.edge_len);
  const int x_s = i * (n + m), x_u = x_s + n;
  const int y_dyn = i * (n + ck.cn), y_nc = y_dyn + n, y_next = y_dyn + n + ck.cn;
  const int y_ec = T * (n + ck.cn) + n + ck.cT + i * ck.ce;
  const int z_n = i * ck.gn, z_e = T * ck.gn + ck.gT + i * ck.ge;
  const double *w = w_all + p * ck.z_dim, *r1 = r1_all + p * xt, *r2 = r2_all + p * ck.y_dim;
  const double *r3 = r3_all + p * ck.z_dim;
  const double *x_x = x_all + p * full, *x_y = x_x + xt, *x_z = x_y + ck.y_dim;
  double *y_x = y_all + p * full, *y_y = y_x + xt, *y_z = y_y + ck.y_dim;

  double *buf = sm, *v = buf + ck.lds_item;
  // vector slices in LDS: x_i | u_i | ydyn_i | ydyn_{i+1} | yc_node | z_node | yc_edge | z_edge
  double *vx = v, *vu = vx + n, *vd = vu + m, *vdn = vd + n, *vyc = vdn + n, *vzn = vyc + c, *vye = vzn + g,
         *vze = vye + ce;
  stage_copy2(buf, item, node_len + edge_len, tid);
  for (int k = tid; k < n; k += TPB) {
    vx[k] = x_x[x_s + k];
    vd[k] = x_y[y_dyn + k];
    if (!last)
      vdn[k] = x_y[y_next + k];
  }
  if (!last)
    for (int k = tid; k < m; k += TPB)
      vu[k] = x_x[x_u + k];
  for (int k = tid; k < c; k += TPB)
    vyc[k] = x_y[y_nc + k];
  for (int k = tid; k < g; k += TPB)
    vzn[k] = x_z[z_n + k];
  for (int k = tid; k < ce; k += TPB)
    vye[k] = x_y[y_ec + k];
  for (int k = tid; k < ge; k += TPB)
    vze[k] = x_z[z_e + k];
  __syncthreads();
  const double *Q = buf, *Jc = buf + nn, *Jg = Jc + c * n, *eb = buf + node_len;
  const double *eQ = eb, *M = eb + nn, *R = M + nm, *A = R + m * m, *B = A + nn, *Jxc = B + nm, *Juc = Jxc + ce * n,
               *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  auto dotr = [](const double *Mx, int ld, int row, int cols, const double *vec) { // row of a column-major block
    double acc = 0.0;
    for (int col = 0; col < cols; ++col)
      acc += Mx[row + ld * col] * vec[col];
    return acc;
  };
  auto dotc = [](const double *Mx, int ld, int col, int rows, const double *vec) { // column
    const double *a = Mx + ld * col;
    double acc = 0.0;
    for (int r = 0; r < rows; ++r)
      acc += a[r] * vec[r];
    return acc;
  };
  const int n_rows = n + (last ? 0 : m + n) + c + g + ce + ge + (i == 0 ? n : 0);
  for (int r = tid; r < n_rows; r += TPB) {
    int q = r;
    if (q < n) { // state rows: H x + C^T y + G^T z + r1 x
      double acc = dotr(Q, n, q, n, vx) + dotc(Jc, c, q, c, vyc) + dotc(Jg, g, q, g, vzn) - vd[q];
      if (!last)
        acc += dotr(eQ, n, q, n, vx) + dotr(M, n, q, m, vu) + dotc(A, n, q, n, vdn) + dotc(Jxc, ce, q, ce, vye) +
               dotc(Jxg, ge, q, ge, vze);
      y_x[x_s + q] += acc + r1[x_s + q] * vx[q];
      continue;
    }
    q -= n;
    if (!last) {
      if (q < m) { // control rows
        const double acc = dotc(M, n, q, n, vx) + dotr(R, m, q, m, vu) + dotc(B, n, q, n, vdn) +
                           dotc(Juc, ce, q, ce, vye) + dotc(Jug, ge, q, ge, vze);
        y_x[x_u + q] += acc + r1[x_u + q] * vu[q];
        continue;
      }
      q -= m;
      if (q < n) { // dynamics rows of node i + 1: A x + B u - x_{i+1} - r2 y
        const double acc = dotr(A, n, q, n, vx) + dotr(B, n, q, m, vu) - x_x[x_s + n + m + q];
        y_y[y_next + q] += acc - r2[y_next + q] * vdn[q];
        continue;
      }
      q -= n;
    }
    if (q < c) {
      y_y[y_nc + q] += dotr(Jc, c, q, n, vx) - r2[y_nc + q] * vyc[q];
      continue;
    }
    q -= c;
    if (q < g) {
      y_z[z_n + q] += dotr(Jg, g, q, n, vx) - (w[z_n + q] + r3[z_n + q]) * vzn[q];
      continue;
    }
    q -= g;
    if (q < ce) {
      y_y[y_ec + q] += dotr(Jxc, ce, q, n, vx) + dotr(Juc, ce, q, m, vu) - r2[y_ec + q] * vye[q];
      continue;
    }
    q -= ce;
    if (q < ge) {
      y_z[z_e + q] += dotr(Jxg, ge, q, n, vx) + dotr(Jug, ge, q, m, vu) - (w[z_e + q] + r3[z_e + q]) * vze[q];
      continue;
    }
    q -= ge; // i == 0: the root's dynamics rows, -x_root - r2 y (helpers.cpp:1081-1085)
    y_y[y_dyn + q] += -vx[q] - r2[y_dyn + q] * vd[q];
  }
}

} // namespace kkt
} // namespace sipamd
