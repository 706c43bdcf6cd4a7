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
  // 1: the Riccati sweep reads ddyn_dx | ddyn_du in place (sip_lqr_factor_solve_split): the stage
  // blocks of mats are [Q_mod | dyn_r2 | M_mod | R_mod] and nothing copies A | B (helpers.cpp:365-366)
  int split;
  // 1 (with split): Q_mod and R_mod leave as packed lower triangles (SIP_LQR_LAYOUT_SYMMETRIC): the tiles below
  // compute the lower triangle anyway -- no mirror writes, a third less of mats written here and read by the sweep
  int sym;
};

typedef double d2_t __attribute__((ext_vector_type(2)));

// The reference's Newton-KKT benchmark family (benchmarks/newton_kkt_benchmark.cpp:59-80): every node has n states,
// every edge m controls and (c, g) = (max(1, n / 2), max(1, 2 m)) constraint rows, the terminal node (c, g), interior
// nodes none.  Kernels instantiated with FN > 0 take every dimension but the horizon from the template: the index
// arithmetic, the loops over constraint rows and most of the branches of the generic code fold away (the
// condensation is issue-bound: ~2 400 instructions per stage in the generic form).  FN = 0: dimensions from `ck`.
template <int FN, int FM>
__device__ __forceinline__ ChainKkt family_dims(ChainKkt ck) {
  if constexpr (FN > 0) {
    constexpr int c = FN / 2 > 0 ? FN / 2 : 1, g = 2 * FM > 0 ? 2 * FM : 1;
    constexpr int n = FN, m = FM;
    ck.n = n, ck.m = m, ck.cn = 0, ck.gn = 0, ck.cT = c, ck.gT = g, ck.ce = c, ck.ge = g;
    ck.node_len = n * n;
    ck.edge_len = 2 * n * n + 2 * n * m + m * m + (c + g) * (n + m);
    ck.vecs_stage = 2 * n + m;
    const int qlen = ck.sym ? n * (n + 1) / 2 : n * n, rlen = ck.sym ? m * (m + 1) / 2 : m * m;
    ck.mats_stage = ck.split ? (qlen + n) + (n * m + rlen) : (n * n + n) + (n * n + 2 * n * m + m * m);
    constexpr int cgn = c + g, cge = c + g;
    ck.lds_item = (n * n + cgn * n + ck.edge_len + 1) / 2 * 2;
    ck.lds_tail = (cgn * n + cge * (n + m) + 1) / 2 * 2;
    ck.lds_rows = (cgn + cge + 1) / 2 * 2;
  }
  return ck;
}

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
// load in turn: one HBM round trip per 1 KB).  The loads are unconditional
// (index clamped to the last piece) and only the LDS stores are predicated:
// predicated loads compile to a branch per load and zero-initialised
// destination registers, whose writes make the compiler drain every load that
// is still in flight (the caller's early reads) before the copy starts.
__device__ __forceinline__ void stage_copy2(double *dst, const double *__restrict__ src, int len, int tid) {
  constexpr int U = 8;
  // 16-byte pieces when both ends allow it (len even, both 16-byte aligned)
  if (((len | (int)((uintptr_t)src >> 3) | (int)((uintptr_t)dst >> 3)) & 1) == 0) {
    const d2_t *s2 = (const d2_t *)src;
    d2_t *d2 = (d2_t *)dst;
    const int len2 = len >> 1;
    if (len2 <= U * TPB) { // one pass, straight-line: a loop's back edge makes the compiler wait for
                           // loads of "the previous iteration", i.e. for everything in flight
      if (len2 > 0) {
        d2_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          v[u] = s2[min(tid + u * TPB, len2 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (tid + u * TPB < len2)
            d2[tid + u * TPB] = v[u];
      }
      return;
    }
    for (int base = tid; base < len2; base += U * TPB) {
      d2_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = s2[min(base + u * TPB, len2 - 1)];
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
        v[u] = src[min(base + u * TPB, len - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (base + u * TPB < len)
          dst[base + u * TPB] = v[u];
    }
  }
}

// Two HBM -> LDS copies with the loads of BOTH in flight before the first LDS store (two calls of
// stage_copy2 would cost two HBM round trips).  Falls back to consecutive copies when a range is
// unaligned or longer than 4 KB.
__device__ __forceinline__ void stage_copy_pair(double *d0, const double *__restrict__ s0, int l0, double *d1,
                                                const double *__restrict__ s1, int l1, int tid) {
  constexpr int U = 4;
  auto vec_ok = [](const double *s, const double *d, int l) {
    return ((l | (int)((uintptr_t)s >> 3) | (int)((uintptr_t)d >> 3)) & 1) == 0 && (l >> 1) <= U * TPB;
  };
  if (l0 > 0 && l1 > 0 && vec_ok(s0, d0, l0) && vec_ok(s1, d1, l1)) {
    const d2_t *a2 = (const d2_t *)s0, *b2 = (const d2_t *)s1;
    d2_t *da = (d2_t *)d0, *db = (d2_t *)d1;
    const int n0 = l0 >> 1, n1 = l1 >> 1;
    d2_t va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      va[u] = a2[min(tid + u * TPB, n0 - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u)
      vb[u] = b2[min(tid + u * TPB, n1 - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (tid + u * TPB < n0)
        da[tid + u * TPB] = va[u];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (tid + u * TPB < n1)
        db[tid + u * TPB] = vb[u];
  } else {
    stage_copy2(d0, s0, l0, tid);
    stage_copy2(d1, s1, l1, tid);
  }
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// ---- condensation of one stage (node i + edge i of problem p), in pieces so that the one-stage
// kernel and the software-pipelined kernel below share them ----
struct CondenseItem {
  long p;
  int i;
  bool last;
  int c, g, ce, ge, nrows, node_len, edge_len;
  int y_dyn, y_nc, y_ec, z_n, z_e; // flattened orderings (types.cpp:24-64) of a uniform chain
  const double *item, *r1, *yinv, *zinv, *b, *b_y, *b_z;
  double *mats, *vecs;
};

template <bool WITH_RHS>
__device__ __forceinline__ CondenseItem condense_item(const ChainKkt &ck, const long p, const int i,
                                                      const double *model_all, const double *r1_all,
                                                      const double *inv_all, double *mats_all, const double *b_all,
                                                      double *vecs_all) {
  CondenseItem it;
  const int n = ck.n, m = ck.m, T = ck.T;
  it.p = p, it.i = i, it.last = i == T;
  it.c = it.last ? ck.cT : ck.cn, it.g = it.last ? ck.gT : ck.gn;
  it.ce = it.last ? 0 : ck.ce, it.ge = it.last ? 0 : ck.ge;
  it.nrows = it.c + it.g + it.ce + it.ge;
  it.node_len = n * n + (it.c + it.g) * n, it.edge_len = it.last ? 0 : ck.edge_len;
  const long kkt = (long)ck.x_dim + ck.y_dim + ck.z_dim;
  it.item = model_all + p * ck.model_len + (long)i * (ck.node_len + ck.edge_len);
  it.r1 = r1_all != nullptr ? r1_all + p * ck.x_dim + i * (n + m) : nullptr;
  it.yinv = inv_all + p * ((long)ck.y_dim + ck.z_dim), it.zinv = it.yinv + ck.y_dim;
  it.y_dyn = i * (n + ck.cn), it.y_nc = it.y_dyn + n, it.y_ec = T * (n + ck.cn) + n + ck.cT + i * ck.ce;
  it.z_n = i * ck.gn, it.z_e = T * ck.gn + ck.gT + i * ck.ge;
  it.b = WITH_RHS ? b_all + p * kkt : nullptr;
  it.b_y = WITH_RHS ? it.b + ck.x_dim : nullptr, it.b_z = WITH_RHS ? it.b_y + ck.y_dim : nullptr;
  it.mats = mats_all + p * ck.mats_len + (long)i * ck.mats_stage;
  it.vecs = WITH_RHS ? vecs_all + p * ck.vecs_len + (long)i * ck.vecs_stage : nullptr;
  return it;
}

// entry of the right-hand side / weight 1/r2, 1/(w+r3) that belongs to constraint row k of the
// stage; rows are ordered [node c | node g | edge c | edge g]
__device__ __forceinline__ double condense_rhs_row(const CondenseItem &it, const int k, const double *by,
                                                   const double *bz) {
  if (k < it.c)
    return by[it.y_nc + k];
  if (k < it.c + it.g)
    return bz[it.z_n + (k - it.c)];
  if (k < it.c + it.g + it.ce)
    return by[it.y_ec + (k - it.c - it.g)];
  return bz[it.z_e + (k - it.c - it.g - it.ce)];
}
__device__ __forceinline__ double condense_weight_row(const CondenseItem &it, const int k) {
  if (k < it.c)
    return it.yinv[it.y_nc + k];
  if (k < it.c + it.g)
    return it.zinv[it.z_n + (k - it.c)];
  if (k < it.c + it.g + it.ce)
    return it.yinv[it.y_ec + (k - it.c - it.g)];
  return it.zinv[it.z_e + (k - it.c - it.g - it.ce)];
}

// The small reads of a stage (weight and right-hand side of the lane's constraint row, r1, dyn_r2,
// the lane's entries of b), held in registers from the moment they are issued.
struct CondensePre {
  double w, b, q, dy, d, r1;
};

template <bool WITH_RHS, bool MATS>
__device__ __forceinline__ void condense_prefetch(const ChainKkt &ck, const CondenseItem &it, const int tid,
                                                  CondensePre &pre) {
  const int n = ck.n, m = ck.m;
  pre.w = pre.b = pre.q = pre.dy = pre.d = pre.r1 = 0.0;
  if (tid < it.nrows) {
    pre.w = condense_weight_row(it, tid);
    if (WITH_RHS)
      pre.b = condense_rhs_row(it, tid, it.b_y, it.b_z);
  }
  if (MATS) {
    if (tid < n)
      pre.d = it.yinv[it.y_dyn + tid];
    if (tid < (it.last ? n : n + m))
      pre.r1 = it.r1[tid];
  }
  if (WITH_RHS) {
    if (tid < n) {
      pre.q = it.b[it.i * (n + m) + tid];
      pre.dy = it.b_y[it.y_dyn + tid];
    } else if (!it.last && tid >= 32 && tid - 32 < m) {
      pre.q = it.b[it.i * (n + m) + n + (tid - 32)];
    }
  }
}

// the LDS side of the small reads: weights | weighted right-hand side rows | r1 of the stage
template <bool WITH_RHS, bool MATS>
__device__ __forceinline__ void condense_commit(const ChainKkt &ck, const CondenseItem &it, const int tid,
                                                const CondensePre &pre, double *wl, double *wr, double *r1s) {
  if (tid < it.nrows) {
    wl[tid] = pre.w;
    if (WITH_RHS)
      wr[tid] = pre.w * pre.b; // weights(constraint) * rhs(constraint), helpers.cpp:143
  }
  for (int k = tid + TPB; k < it.nrows; k += TPB) { // more than 64 constraint rows in a stage
    const double wk = condense_weight_row(it, k);
    wl[k] = wk;
    if (WITH_RHS)
      wr[k] = wk * condense_rhs_row(it, k, it.b_y, it.b_z);
  }
  if (MATS && tid < (it.last ? ck.n : ck.n + ck.m))
    r1s[tid] = pre.r1;
}

// Everything after the stage is in LDS (`buf`: the stage's model image; for MATS = false only its
// Jacobians, at their usual places).
template <bool WITH_RHS, bool MATS>
__device__ __forceinline__ void condense_compute(const ChainKkt &ck, const CondenseItem &it, const int tid,
                                                 const CondensePre &pre, const double *buf, const double *wl,
                                                 double *wr, const double *r1s, const int ncols_rhs,
                                                 const long b_col_stride, const long vecs_col_stride,
                                                 double *obuf = nullptr) {
  const int n = ck.n, m = ck.m, nn = n * n, nm = n * m;
  const bool last = it.last;
  const int c = it.c, g = it.g, ce = it.ce, ge = it.ge;
  // MATS: `buf` is the whole stage image.  Otherwise (right-hand sides only) it holds the constraint Jacobians alone,
  // node's then edge's, packed: a fifth of the LDS, three times the workgroups per CU
  const double *Jc = MATS ? buf + nn : buf, *Jg = Jc + c * n;
  const double *eb = buf + it.node_len; // edge item (MATS)
  const int o_m = nn, o_r = o_m + nm, o_a = o_r + m * m, o_b = o_a + nn, o_j = o_b + nm;
  const double *Jxc = MATS ? eb + o_j : buf + (c + g) * n, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  // MATS: the stage block of mats is assembled in LDS (obuf) and leaves as one coalesced copy -- the tile
  // epilogue below scatters single scalars (lane l of a tile row writes column l: 96-byte strides), which as
  // global stores were one 8-byte L2 transaction per lane and element
  double *mats = MATS ? obuf : it.mats;

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
    const bool sym = ck.sym != 0; // (only with split)
    const int qlen = sym ? n * (n + 1) / 2 : nn;
    double *Qm = mats, *Am = mats + qlen + n, *Bm = Am + nn, *Mm = ck.split ? Am : Bm + nm, *Rm = Mm + nm;
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
                v += r1s[I];
              if (!last)
                v += eb[I + n * J];
              v += t;
              if (sym) {
                Qm[J * n - (J * (J - 1)) / 2 + (I - J)] = v; // packed column J, rows J .. n-1
              } else {
                Qm[I + n * J] = v;
                Qm[J + n * I] = v;
              }
            }
          } else if (I < ncols) {
            const int u = I - n;
            if (J < n) { // M_mod(x_J, u), :341-352
              Mm[J + n * u] = eb[o_m + J + n * u] + t;
            } else if (J - n <= u) { // R_mod, lower, mirrored (:344-354)
              const int uj = J - n;
              double v = eb[o_r + u + m * uj];
              if (u == uj)
                v += r1s[n + u];
              v += t;
              if (sym) {
                Rm[uj * m - (uj * (uj - 1)) / 2 + (u - uj)] = v;
              } else {
                Rm[u + m * uj] = v;
                Rm[uj + m * u] = v;
              }
            }
          }
        }
      }
    }
    if (tid < n)
      mats[(ck.sym ? n * (n + 1) / 2 : nn) + tid] = pre.d; // dyn_r2
    if (!last && !ck.split) {
      for (int k = tid; k < nn; k += TPB) // ddyn_dx, ddyn_du, :365-366
        Am[k] = eb[o_a + k];
      for (int k = tid; k < nm; k += TPB)
        Bm[k] = eb[o_b + k];
    }
  }
  if (MATS) {
    __syncthreads();
    const int len = last ? (ck.sym ? n * (n + 1) / 2 : nn) + n : ck.mats_stage;
    for (int k = tid; k < len; k += TPB)
      it.mats[k] = obuf[k];
  }
  if (WITH_RHS && !MATS && ncols_rhs > 1) {
    // Several right-hand sides (the columns of J_theta, helpers.cpp:414-747) in TWO phases for all of
    // them -- a loop over the columns exposed two barriers and two dependent global reads per column:
    // (1) every column's weighted rows, wr[col][k] = w_k * rhs_col(k): all (col, k) pairs at once;
    // (2) every (col, state row / control row) pair: one lane each, the same sums in the same order
    //     as the single-column code below (bitwise the same numbers).
    const int i = it.i, R = ck.lds_rows, nrows = it.nrows, per = n + m;
    constexpr int U = 4; // pairs per lane and trip: their global reads are all requested first
    for (int e0 = tid; e0 < ncols_rhs * nrows; e0 += U * TPB) {
      double bv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * TPB, col = e / nrows, k = e - col * nrows;
        const double *bc_y = it.b + col * b_col_stride + ck.x_dim, *bc_z = bc_y + ck.y_dim;
        bv[u] = e < ncols_rhs * nrows ? condense_rhs_row(it, k, bc_y, bc_z) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * TPB, col = e / nrows, k = e - col * nrows;
        if (e < ncols_rhs * nrows)
          wr[col * R + k] = wl[k] * bv[u];
      }
    }
    __syncthreads();
    for (int e0 = tid; e0 < ncols_rhs * per; e0 += 2 * TPB) {
      double q0[2], dy[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * TPB, col = e / per, j = e - col * per;
        q0[u] = dy[u] = 0.0;
        if (e < ncols_rhs * per && !(last && j >= n)) {
          const double *bc = it.b + col * b_col_stride, *bc_y = bc + ck.x_dim;
          q0[u] = bc[i * (n + m) + j];
          if (j < n)
            dy[u] = bc_y[it.y_dyn + j];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * TPB, col = e / per, j = e - col * per;
        if (e >= ncols_rhs * per || (last && j >= n))
          continue;
        double *vc = it.vecs + col * vecs_col_stride;
        const double *wr_n = wr + col * R, *wr_e = wr_n + c + g;
        double acc = -q0[u];
        if (j < n) {
          acc = dot_seq<true>(acc, Jc + c * j, wr_n, c);
          acc = dot_seq<true>(acc, Jg + g * j, wr_n + c, g);
          if (!last) {
            acc = dot_seq<true>(acc, Jxc + ce * j, wr_e, ce);
            acc = dot_seq<true>(acc, Jxg + ge * j, wr_e + ce, ge);
          }
          vc[j] = acc;
          vc[n + j] = -dy[u];
        } else {
          const int d = j - n;
          acc = dot_seq<true>(acc, Juc + ce * d, wr_e, ce);
          acc = dot_seq<true>(acc, Jug + ge * d, wr_e + ce, ge);
          vc[2 * n + d] = acc;
        }
      }
    }
  } else if (WITH_RHS) { // q_mod, c_mod, r_mod (helpers.cpp:752-812)
    const double *wr_n = wr, *wr_e = wr + c + g;
    const int i = it.i;
    for (int col = 0; col < ncols_rhs; ++col) {
      const double *bc = it.b + col * b_col_stride, *bc_y = bc + ck.x_dim, *bc_z = bc_y + ck.y_dim;
      double *vc = it.vecs + col * vecs_col_stride;
      if (col > 0) { // weighted right-hand side rows of this column
        __syncthreads();
        for (int k = tid; k < it.nrows; k += TPB)
          wr[k] = wl[k] * condense_rhs_row(it, k, bc_y, bc_z);
        __syncthreads();
      }
      if (tid < n) {
        double acc = -(col == 0 ? pre.q : bc[i * (n + m) + tid]);
        acc = dot_seq<true>(acc, Jc + c * tid, wr_n, c);
        acc = dot_seq<true>(acc, Jg + g * tid, wr_n + c, g);
        if (!last) {
          acc = dot_seq<true>(acc, Jxc + ce * tid, wr_e, ce);
          acc = dot_seq<true>(acc, Jxg + ge * tid, wr_e + ce, ge);
        }
        vc[tid] = acc;
        vc[n + tid] = -(col == 0 ? pre.dy : bc_y[it.y_dyn + tid]);
      } else if (!last && tid >= 32 && tid - 32 < m) { // another half of the wave: r_mod
        const int d = tid - 32;
        double acc = -(col == 0 ? pre.q : bc[i * (n + m) + n + d]);
        acc = dot_seq<true>(acc, Juc + ce * d, wr_e, ce);
        acc = dot_seq<true>(acc, Jug + ge * d, wr_e + ce, ge);
        vc[2 * n + d] = acc;
      }
    }
  }
}

// WITH_RHS: also q_mod / r_mod / c_mod from b.  MATS = false (the split solve path): ONLY those --
// the Jacobians of the stage are staged, no Q_mod / M_mod / R_mod / A / B work, `status` problems
// with a failed factorization are skipped.
// (the same for the rhs-only condensation of sip_kkt_solve: 8 wavefronts per SIMD, solve 0.63 -> 0.557 ms; the K x
// kernel loses with any cap -- 0.48 -> 0.49 / 0.72 ms at 6 / 8 -- and keeps its 96 registers)
#ifndef SIP_KKT_RHS_WAVES
#define SIP_KKT_RHS_WAVES 8
#endif
template <bool WITH_RHS, bool MATS = true, int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FN > 0 && !MATS ? SIP_KKT_RHS_WAVES : 1, 8)))
condense_chain_kernel(const ChainKkt ck_in, const double *__restrict__ model_all, const double *__restrict__ r1_all,
                      const double *__restrict__ inv_all, double *__restrict__ mats_all,
                      const double *__restrict__ b_all, double *__restrict__ vecs_all, long batch,
                      const int32_t *__restrict__ status = nullptr, const int ncols = 1,
                      const long b_col_stride = 0, const long vecs_col_stride = 0) {
  // ncols > 1 (the columns of J_theta, helpers.cpp:414-747): column j's right-hand side is
  // b_all + j * b_col_stride, its q_mod / r_mod / c_mod go to vecs_all + j * vecs_col_stride; the
  // Jacobians of the stage are staged once for all columns.
  static_assert(MATS || WITH_RHS, "nothing to do");
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  // wr: one block of lds_rows per right-hand-side column when there are several (MATS = false)
  double *buf = sm, *wl = buf + (MATS ? ck.lds_item : ck.lds_tail), *wr = wl + ck.lds_rows,
         *r1s = wr + (long)(MATS || ncols < 1 ? 1 : ncols) * ck.lds_rows;
  double *obuf = r1s + ((ck.n + ck.m + 1) & ~1); // MATS: the stage block of mats before it leaves
  const long p = blockIdx.x / (ck.T + 1);
  const int i = blockIdx.x - (unsigned)(p * (ck.T + 1));
  if (p >= batch || (!MATS && status != nullptr && status[p] != 0))
    return;
  const int tid = threadIdx.x;
  const CondenseItem it = condense_item<WITH_RHS>(ck, p, i, model_all, r1_all, inv_all, mats_all, b_all, vecs_all);
  // Every small read of the wavefront is issued AHEAD of the stage copy: one HBM round trip per
  // wavefront, not three.
  CondensePre pre;
  condense_prefetch<WITH_RHS, MATS>(ck, it, tid, pre);
  // whole stage (node item + edge item are adjacent in the model arena) -> LDS
  if (MATS) {
    stage_copy2(buf, it.item, it.node_len + it.edge_len, tid);
  } else { // the constraint Jacobians only, packed (node's, then edge's)
    const int n = ck.n, m = ck.m, nn = n * n;
    const int o_tail = it.node_len + 2 * nn + 2 * n * m + m * m;
    if (it.last)
      stage_copy2(buf, it.item + nn, (it.c + it.g) * n, tid);
    else
      stage_copy_pair(buf, it.item + nn, (it.c + it.g) * n, buf + (it.c + it.g) * n, it.item + o_tail,
                      (it.ce + it.ge) * (n + m), tid);
  }
  condense_commit<WITH_RHS, MATS>(ck, it, tid, pre, wl, wr, r1s);
  __syncthreads();
  condense_compute<WITH_RHS, MATS>(ck, it, tid, pre, buf, wl, wr, r1s, ncols, b_col_stride, vecs_col_stride, obuf);
}

// The same condensation, software-pipelined: a wavefront walks `per_block` consecutive stages and
// has the NEXT stage's loads (its model image, 16 bytes per lane and piece, and the small reads) in
// flight in registers while it computes the current one from LDS -- with one stage per wavefront
// the load, compute and store phases of a wavefront do not overlap and the kernel ran at
// load time + compute time + store time (245 + 180 + 110 us at the f1 shape).  Requires the stage
// image to be one pass of 16-byte pieces: (node_len + edge_len) / 2 <= PIPE_U * 64, even lengths,
// 16-byte aligned arenas (checked on the host).
// (Three wavefronts per SIMD -- what its 156 registers give -- is the best occupancy for it: capped to 2, 4 or 5 the
// Newton-KKT step took 1.06 / 0.95 / 1.42 ms instead of 0.92.)
#ifdef SIP_KKT_STAMPS // diagnostic build (tools/kkt_ab_build.sh ... -DSIP_KKT_STAMPS): cycles per segment, summed over wavefronts
__device__ unsigned long long g_kkt_seg[16];
#define KKT_SEG(k)                                                                                   \
  do {                                                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                    \
    seg[k] += now_ - seg_last;                                                                       \
    seg_last = now_;                                                                                 \
  } while (0)
#else
#define KKT_SEG(k)                                                                                   \
  do {                                                                                               \
  } while (0)
#endif
constexpr int PIPE_U = 8;
// (no cap: held to 5 wavefronts per SIMD -- 96 registers -- the family instantiation of the pipelined condensation
// makes the step 0.85 ms against 0.77)
#ifndef SIP_KKT_PIPE_WAVES
#define SIP_KKT_PIPE_WAVES 1
#endif
template <bool WITH_RHS, int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FN > 0 ? SIP_KKT_PIPE_WAVES : 1, 8)))
condense_chain_pipe_kernel(const ChainKkt ck_in, const double *__restrict__ model_all,
                           const double *__restrict__ r1_all, const double *__restrict__ inv_all,
                           double *__restrict__ mats_all, const double *__restrict__ b_all,
                           double *__restrict__ vecs_all, const long batch, const int per_block) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  double *buf = sm, *wl = buf + ck.lds_item, *wr = wl + ck.lds_rows, *r1s = wr + ck.lds_rows;
  double *obuf = r1s + ((ck.n + ck.m + 1) & ~1);
  const int tid = threadIdx.x;
  const long total = batch * (ck.T + 1);
  const long first = (long)blockIdx.x * per_block;
  const long end = first + per_block < total ? first + per_block : total;
  if (first >= end)
    return;
  // (problem, stage) of the walk: one 64-bit division for the first item, then counted on
  long walk_p = first / (ck.T + 1);
  int walk_i = (int)(first - walk_p * (ck.T + 1));
  auto item_here = [&]() {
    return condense_item<WITH_RHS>(ck, walk_p, walk_i, model_all, r1_all, inv_all, mats_all, b_all, vecs_all);
  };
  auto step_walk = [&]() {
    if (++walk_i > ck.T)
      walk_i = 0, ++walk_p;
  };
  // split: the pieces of ddyn_dx | ddyn_du are not fetched (nothing here reads them; the Riccati
  // sweep takes them from the arena itself) -- their lanes re-read the piece in front of them
  // (whole pieces inside the block only: with odd m it starts on an odd scalar, and the piece across its first
  // boundary carries the last entry of R)
  const int ab_at = ck.node_len + ck.n * ck.n + ck.n * ck.m + ck.m * ck.m;
  const int ab_first = ck.split ? (ab_at + 1) >> 1 : 0;
  const int ab_end = ck.split ? (ab_at + ck.n * ck.n + ck.n * ck.m) >> 1 : 0;
  auto image_load = [&](const CondenseItem &it, d2_t (&v)[PIPE_U]) {
    // (an item of odd length, or an arena at an odd scalar: 16-byte pieces from 8-byte aligned addresses; the last
    // piece of an odd item ends with the first scalar of the item behind it)
    typedef d2_t d2u_t __attribute__((aligned(8)));
    const d2u_t *s2 = (const d2u_t *)it.item;
    const int len2 = (it.node_len + it.edge_len + 1) >> 1;
#pragma unroll
    for (int u = 0; u < PIPE_U; ++u) {
      int q = min(tid + u * TPB, len2 - 1);
      if (!it.last && q >= ab_first && q < ab_end)
        q = ab_first - 1;
      v[u] = s2[q];
    }
  };
#ifdef SIP_KKT_STAMPS
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, seg_last = __builtin_amdgcn_s_memtime();
#endif
  CondenseItem cur = item_here();
  CondensePre pre;
  d2_t img[PIPE_U];
  condense_prefetch<WITH_RHS, true>(ck, cur, tid, pre);
  image_load(cur, img);
  KKT_SEG(0);
  for (long idx = first; idx < end; ++idx) {
    { // registers -> LDS
      d2_t *d2 = (d2_t *)buf;
      const int len2 = (cur.node_len + cur.edge_len + 1) >> 1;
#pragma unroll
      for (int u = 0; u < PIPE_U; ++u) // (lanes past the end hold a copy of the last piece: the same store)
        d2[min(tid + u * TPB, len2 - 1)] = img[u];
    }
    KKT_SEG(1); // wait for the image + its stores to LDS
    condense_commit<WITH_RHS, true>(ck, cur, tid, pre, wl, wr, r1s);
    __syncthreads();
    KKT_SEG(2);
    const CondensePre now = pre;
    const CondenseItem nowit = cur;
    if (idx + 1 < end) { // the next stage's loads fly during this stage's compute
      step_walk();
      cur = item_here();
      condense_prefetch<WITH_RHS, true>(ck, cur, tid, pre);
      image_load(cur, img);
    }
    KKT_SEG(3); // issue of the next stage's loads
    condense_compute<WITH_RHS, true>(ck, nowit, tid, now, buf, wl, wr, r1s, 1, 0, 0, obuf);
    KKT_SEG(4);
    __syncthreads(); // every reader of the LDS image is done before it is overwritten
    KKT_SEG(5);
  }
#ifdef SIP_KKT_STAMPS
  if (tid == 0)
    for (int k = 0; k < 8; ++k)
      atomicAdd(&g_kkt_seg[k], seg[k]);
#endif
}

// x, u, y scatter + multipliers of node i and edge i (helpers.cpp:817-892).  COLS: the instantiation
// for several columns per launch (the two-phase form below) -- a kernel of its own so that the
// single-column one keeps its 58 registers and eight wavefronts per SIMD.
// Wavefronts per SIMD the register allocator is held to in the benchmark-family instantiation of the single-column
// recovery: unrolled over the compile-time dimensions it took 122 registers (4 wavefronts per SIMD) where the generic one
// takes 58 (8); held to 64 registers the Newton-KKT step runs 0.768 against 0.792 ms (6 wavefronts: 0.787).
#ifndef SIP_KKT_RECOVER_WAVES
#define SIP_KKT_RECOVER_WAVES 8
#endif
template <bool COLS = false, int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FN > 0 && !COLS ? SIP_KKT_RECOVER_WAVES : 1, 8)))
recover_chain_kernel(const ChainKkt ck_in, const double *__restrict__ model_all, const double *__restrict__ b_all,
                     const double *__restrict__ inv_all, const double *__restrict__ lqr_sol_all,
                     double *__restrict__ sol_all, const int32_t *__restrict__ status, long batch,
                     const int ncols = 1, const long b_col_stride = 0, const long lqr_col_stride = 0,
                     const long sol_col_stride = 0) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
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
  // column 0's small reads (stagewise solution, right-hand side and weight of the lane's constraint
  // row) are issued ahead of the stage copy: one HBM round trip per wavefront instead of three
  double pre_x = 0.0, pre_y = 0.0, pre_bn = 0.0, pre_wn = 0.0, pre_be = 0.0, pre_we = 0.0;
  {
    const double *b_y = b_all + p * kkt + ck.x_dim, *b_z = b_y + ck.y_dim;
    const double *ls = lqr_sol_all + p * ck.vecs_len + (long)i * ck.vecs_stage;
    if (tid < n) {
      pre_x = ls[tid];
      pre_y = ls[n + tid];
    } else if (!last && tid >= 32 && tid - 32 < m) {
      pre_x = ls[2 * n + (tid - 32)];
    }
    if (tid < c)
      pre_bn = b_y[y_nc + tid], pre_wn = yinv[y_nc + tid];
    else if (tid < c + g)
      pre_bn = b_z[z_n + (tid - c)], pre_wn = zinv[z_n + (tid - c)];
    if (tid < ce)
      pre_be = b_y[y_ec + tid], pre_we = yinv[y_ec + tid];
    else if (tid < ce + ge)
      pre_be = b_z[z_e + (tid - ce)], pre_we = zinv[z_e + (tid - ce)];
  }
  if (last) // [dc_dx | dg_dx] of the node, [dc_dx | dc_du | dg_dx | dg_du] of the edge
    stage_copy2(jn, item + nn, (c + g) * n, tid);
  else
    stage_copy_pair(jn, item + nn, (c + g) * n, je, item + nn + (c + g) * n + 2 * nn + 2 * nm + m * m,
                    (ce + ge) * (n + m), tid);
  const double *Jxc = je, *Juc = Jxc + ce * n, *Jxg = Juc + ce * m, *Jug = Jxg + ge * n;
  if constexpr (COLS) {
    // The columns of K^-1 J_theta (helpers.cpp:414-747) in two phases for all of them, not a loop
    // over the columns (two barriers and two dependent global reads per column): (1) x_i | u_i of
    // every column into LDS (xs: one block of n + m per column) and out to sol; (2) every
    // (column, constraint row) pair, one lane each -- the sums of the single-column code.
    const int per = n + m, rows_n = c + g, rows = rows_n + ce + ge;
    for (int e0 = tid; e0 < ncols * per; e0 += 2 * TPB) { // two pairs per lane and trip, loads first
      double v0[2], v1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * TPB, col = e / per, j = e - col * per;
        v0[u] = v1[u] = 0.0;
        if (e < ncols * per && !(last && j >= n)) {
          const double *ls = lqr_sol_all + col * lqr_col_stride + p * ck.vecs_len + (long)i * ck.vecs_stage; // x_i | y_i | u_i
          v0[u] = j < n ? ls[j] : ls[2 * n + (j - n)];
          if (j < n)
            v1[u] = ls[n + j];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = e0 + u * TPB, col = e / per, j = e - col * per;
        if (e >= ncols * per || (last && j >= n))
          continue;
        double *sol = sol_all + col * sol_col_stride + p * kkt, *sol_y = sol + ck.x_dim;
        xs[col * per + j] = v0[u]; // us of a column sits behind its xs
        sol[i * (n + m) + j] = v0[u];
        if (j < n)
          sol_y[y_dyn + j] = v1[u];
      }
    }
    __syncthreads(); // x_i | u_i of the columns and the Jacobians are in LDS
    // two pairs per lane and trip: their right-hand-side entries and weights are all requested
    // before the first sum starts (one memory round trip per trip, not one per pair)
    constexpr int U = 2;
    for (int e0 = tid; e0 < ncols * rows; e0 += U * TPB) {
      double bv[U], wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * TPB;
        bv[u] = wv[u] = 0.0;
        if (e < ncols * rows) {
          const int col = e / rows, k = e - col * rows;
          const double *b_y = b_all + col * b_col_stride + p * kkt + ck.x_dim, *b_z = b_y + ck.y_dim;
          if (k < c)
            bv[u] = b_y[y_nc + k], wv[u] = yinv[y_nc + k];
          else if (k < rows_n)
            bv[u] = b_z[z_n + (k - c)], wv[u] = zinv[z_n + (k - c)];
          else if (k < rows_n + ce)
            bv[u] = b_y[y_ec + (k - rows_n)], wv[u] = yinv[y_ec + (k - rows_n)];
          else
            bv[u] = b_z[z_e + (k - rows_n - ce)], wv[u] = zinv[z_e + (k - rows_n - ce)];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * TPB;
        if (e >= ncols * rows)
          break;
        const int col = e / rows, k = e - col * rows;
        double *sol_y = sol_all + col * sol_col_stride + p * kkt + ck.x_dim, *sol_z = sol_y + ck.y_dim;
        const double *xc = xs + col * per, *uc = xc + n;
        if (k < c) {
          sol_y[y_nc + k] = (row_dot(jn, k, c, n, xc) - bv[u]) * wv[u];
        } else if (k < rows_n) {
          const int kk = k - c;
          sol_z[z_n + kk] = (row_dot(jn + c * n, kk, g, n, xc) - bv[u]) * wv[u];
        } else if (k < rows_n + ce) {
          const int kk = k - rows_n;
          const double jx = row_dot(Jxc, kk, ce, n, xc), ju = row_dot(Juc, kk, ce, m, uc);
          sol_y[y_ec + kk] = ((jx + ju) - bv[u]) * wv[u];
        } else {
          const int kk = k - rows_n - ce;
          const double jx = row_dot(Jxg, kk, ge, n, xc), ju = row_dot(Jug, kk, ge, m, uc);
          sol_z[z_e + kk] = ((jx + ju) - bv[u]) * wv[u];
        }
      }
    }
    return;
  } else {
  for (int col = 0; col < ncols; ++col) {
    const double *b_y = b_all + col * b_col_stride + p * kkt + ck.x_dim, *b_z = b_y + ck.y_dim;
    const double *ls = lqr_sol_all + col * lqr_col_stride + p * ck.vecs_len + (long)i * ck.vecs_stage; // x_i | y_i | u_i
    double *sol = sol_all + col * sol_col_stride + p * kkt, *sol_y = sol + ck.x_dim, *sol_z = sol_y + ck.y_dim;
    if (col > 0)
      __syncthreads(); // the previous column's readers of xs / us are done
    if (tid < n) {
      const double xv = col == 0 ? pre_x : ls[tid];
      xs[tid] = xv;
      sol[i * (n + m) + tid] = xv;
      sol_y[y_dyn + tid] = col == 0 ? pre_y : ls[n + tid];
    } else if (!last && tid >= 32 && tid - 32 < m) {
      const double uv = col == 0 ? pre_x : ls[2 * n + (tid - 32)];
      us[tid - 32] = uv;
      sol[i * (n + m) + n + (tid - 32)] = uv;
    }
    __syncthreads();
    for (int k = tid; k < c + g; k += TPB) {
      const bool pre = col == 0 && k == tid;
      if (k < c)
        sol_y[y_nc + k] = (row_dot(jn, k, c, n, xs) - (pre ? pre_bn : b_y[y_nc + k])) * (pre ? pre_wn : yinv[y_nc + k]);
      else
        sol_z[z_n + (k - c)] = (row_dot(jn + c * n, k - c, g, n, xs) - (pre ? pre_bn : b_z[z_n + (k - c)])) *
                               (pre ? pre_wn : zinv[z_n + (k - c)]);
    }
    for (int k = tid; k < ce + ge; k += TPB) {
      const bool pre = col == 0 && k == tid;
      if (k < ce) {
        const double jx = row_dot(Jxc, k, ce, n, xs), ju = row_dot(Juc, k, ce, m, us);
        sol_y[y_ec + k] = ((jx + ju) - (pre ? pre_be : b_y[y_ec + k])) * (pre ? pre_we : yinv[y_ec + k]);
      } else {
        const int kk = k - ce;
        const double jx = row_dot(Jxg, kk, ge, n, xs), ju = row_dot(Jug, kk, ge, m, us);
        sol_z[z_e + kk] = ((jx + ju) - (pre ? pre_be : b_z[z_e + kk])) * (pre ? pre_we : zinv[z_e + kk]);
      }
    }
  }
  } // !COLS
}

// y += K x (helpers.cpp:953-1368), or the selected blocks of it (ApplyIO::parts: the five operators
// of helpers.hpp:20-24), for a uniform chain.  The workgroup of stage i owns the state and control
// rows of stage i, the dynamics rows of node i + 1 (plus the root's at i = 0) and the constraint
// rows of node i and edge i: everything they touch is in stage i's model item (staged in LDS) and
// in a few slices of x.  x-space vectors are [x | theta (th entries)].
#ifndef SIP_KKT_APPLY_WAVES
#define SIP_KKT_APPLY_WAVES 1
#endif
template <int FN = 0, int FM = 0>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FN > 0 ? SIP_KKT_APPLY_WAVES : 1, 8)))
apply_chain_kernel(const ChainKkt ck_in, const int th, const double *__restrict__ model_all,
                   const double *__restrict__ w_all, const double *__restrict__ r1_all,
                   const double *__restrict__ r2_all, const double *__restrict__ r3_all, const ApplyIO io,
                   long batch) {
  const ChainKkt ck = family_dims<FN, FM>(ck_in);
  extern __shared__ double sm[];
  const int n = ck.n, m = ck.m, T = ck.T;
  const long p = blockIdx.x / (T + 1);
  const int i = blockIdx.x - (unsigned)(p * (T + 1));
  if (p >= batch)
    return;
  const int tid = threadIdx.x;
  const int parts = io.parts;
  const bool pH = parts & AP_H, pC = parts & AP_C, pCT = parts & AP_CT, pG = parts & AP_G, pGT = parts & AP_GT,
             pR = parts & AP_REG;
  const bool out_x = pH || pCT || pGT || pR, out_y = pC || pR, out_z = pG || pR;
  const bool last = i == T;
  const int c = last ? ck.cT : ck.cn, g = last ? ck.gT : ck.gn;
  const int ce = last ? 0 : ck.ce, ge = last ? 0 : ck.ge;
  const int nn = n * n, nm = n * m;
  const int node_len = nn + (c + g) * n, edge_len = last ? 0 : ck.edge_len;
  const int xt = ck.x_dim + th;
  const double *item = model_all + p * ck.model_len + (long)i * (ck.node_len + ck.edge_len);
  const int x_s = i * (n + m), x_u = x_s + n;
  const int y_dyn = i * (n + ck.cn), y_nc = y_dyn + n, y_next = y_dyn + n + ck.cn;
  const int y_ec = T * (n + ck.cn) + n + ck.cT + i * ck.ce;
  const int z_n = i * ck.gn, z_e = T * ck.gn + ck.gT + i * ck.ge;
  const double *w = w_all + p * ck.z_dim, *r1 = r1_all + p * xt, *r2 = r2_all + p * ck.y_dim;
  const double *r3 = r3_all + p * ck.z_dim; // dereferenced under AP_REG only
  const bool in_x = io.x_x != nullptr, in_y = io.x_y != nullptr, in_z = io.x_z != nullptr;
  const double *x_x = io.x_x + p * io.sx, *x_y = io.x_y + p * io.sy, *x_z = io.x_z + p * io.sz;
  double *y_x = io.y_x + p * io.sx, *y_y = io.y_y + p * io.sy, *y_z = io.y_z + p * io.sz;

  double *buf = sm, *v = buf + ck.lds_item;
  // vector slices in LDS: x_i | u_i | ydyn_i | ydyn_{i+1} | yc_node | z_node | yc_edge | z_edge
  double *vx = v, *vu = vx + n, *vd = vu + m, *vdn = vd + n, *vyc = vdn + n, *vzn = vyc + c, *vye = vzn + g,
         *vze = vye + ce;
  // Output row r of the stage: where it lives in y (null: its space is not an output of this
  // launch), its regularization coefficient (r1 | r2 | w + r3) and, for a dynamics row of node
  // i + 1, the -x_{i+1} entry.
  const int n_rows = n + (last ? 0 : m + n) + c + g + ce + ge + (i == 0 ? n : 0);
  auto row_io = [&](int q, double *&dst, double &coef, double &extra) {
    extra = 0.0, coef = 0.0, dst = nullptr;
    if (q < n) {
      if (out_x)
        dst = y_x + x_s + q, coef = pR ? r1[x_s + q] : 0.0;
      return;
    }
    q -= n;
    if (!last) {
      if (q < m) {
        if (out_x)
          dst = y_x + x_u + q, coef = pR ? r1[x_u + q] : 0.0;
        return;
      }
      q -= m;
      if (q < n) {
        if (out_y)
          dst = y_y + y_next + q, coef = pR ? r2[y_next + q] : 0.0, extra = pC ? x_x[x_s + n + m + q] : 0.0;
        return;
      }
      q -= n;
    }
    if (q < c) {
      if (out_y)
        dst = y_y + y_nc + q, coef = pR ? r2[y_nc + q] : 0.0;
      return;
    }
    q -= c;
    if (q < g) {
      if (out_z)
        dst = y_z + z_n + q, coef = pR ? w[z_n + q] + r3[z_n + q] : 0.0;
      return;
    }
    q -= g;
    if (q < ce) {
      if (out_y)
        dst = y_y + y_ec + q, coef = pR ? r2[y_ec + q] : 0.0;
      return;
    }
    q -= ce;
    if (q < ge) {
      if (out_z)
        dst = y_z + z_e + q, coef = pR ? w[z_e + q] + r3[z_e + q] : 0.0;
      return;
    }
    q -= ge;
    if (out_y)
      dst = y_y + y_dyn + q, coef = pR ? r2[y_dyn + q] : 0.0;
  };
  // Every small read of the wavefront -- the slices of x and the lane's first output row -- is
  // issued ahead of the stage copy: one HBM round trip per wavefront instead of three.
  double *dst0 = nullptr;
  double coef0 = 0.0, extra0 = 0.0, yold0 = 0.0;
  if (tid < n_rows) {
    row_io(tid, dst0, coef0, extra0);
    if (dst0 != nullptr)
      yold0 = *dst0;
  }
  double px = 0.0, pd = 0.0, pdn = 0.0, pu = 0.0, pyc = 0.0, pzn = 0.0, pye = 0.0, pze = 0.0;
  if (tid < n) {
    if (in_x)
      px = x_x[x_s + tid];
    if (in_y) {
      pd = x_y[y_dyn + tid];
      if (!last)
        pdn = x_y[y_next + tid];
    }
  }
  if (!last && tid < m && in_x)
    pu = x_x[x_u + tid];
  if (tid < c && in_y)
    pyc = x_y[y_nc + tid];
  if (tid < g && in_z)
    pzn = x_z[z_n + tid];
  if (tid < ce && in_y)
    pye = x_y[y_ec + tid];
  if (tid < ge && in_z)
    pze = x_z[z_e + tid];
  { // only what the selected blocks read travels: the hull of their ranges in the stage image (H: Q and the edge's
    // Q | M | R; C, CT: dc_dx, A | B, the edge's dc_dx | dc_du; G, GT: the dg blocks) -- SIP is handed the five operators
    // one by one (sip_optimal_control.cpp:147-190), and each staged the whole 6 KB stage
    const int total = node_len + edge_len, o_a = node_len + nn + nm + m * m, o_j = o_a + nn + nm;
    int lo = total, hi = 0;
    auto need = [&](const int a, const int b) {
      if (b > a) {
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
      }
    };
    if (pH) {
      need(0, nn);
      if (!last)
        need(node_len, o_a);
    }
    if (pC || pCT) {
      need(nn, nn + c * n);
      if (!last) {
        need(o_a, o_j);
        need(o_j, o_j + ce * (n + m));
      }
    }
    if (pG || pGT) {
      need(nn + c * n, node_len);
      if (!last)
        need(o_j + ce * (n + m), total);
    }
#ifdef SIP_KKT_APPLY_FULL_STAGE // diagnostic: the whole stage image whatever the mask (tools/kkt_ab_build.sh)
    lo = 0, hi = total;
#endif
    lo &= ~1; // (whole 16-byte pieces where the image allows them)
    hi = (hi + 1) & ~1;
    hi = hi < total ? hi : total;
    if (hi > lo)
      stage_copy2(buf + lo, item + lo, hi - lo, tid);
  }
  if (tid < n) {
    vx[tid] = px;
    vd[tid] = pd;
    if (!last)
      vdn[tid] = pdn;
  }
  if (!last && tid < m)
    vu[tid] = pu;
  if (tid < c)
    vyc[tid] = pyc;
  if (tid < g)
    vzn[tid] = pzn;
  if (tid < ce)
    vye[tid] = pye;
  if (tid < ge)
    vze[tid] = pze;
  for (int k = tid + TPB; k < c + g + ce + ge; k += TPB) { // more than 64 rows of one kind
    if (k < c)
      vyc[k] = in_y ? x_y[y_nc + k] : 0.0;
    if (k < g)
      vzn[k] = in_z ? x_z[z_n + k] : 0.0;
    if (k < ce)
      vye[k] = in_y ? x_y[y_ec + k] : 0.0;
    if (k < ge)
      vze[k] = in_z ? x_z[z_e + k] : 0.0;
  }
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
  const bool all = parts == AP_ALL;
  for (int r = tid; r < n_rows; r += TPB) {
    int q = r;
    double *dst = dst0;
    double coef = coef0, extra = extra0, yold = yold0;
    if (r != tid) {
      row_io(r, dst, coef, extra);
      if (dst != nullptr)
        yold = *dst;
    }
    if (dst == nullptr) // a row of a space this launch does not write
      continue;
    if (q < n) { // state rows: H x + C^T y + G^T z + r1 x
      double acc;
      if (all) { // the whole operator, summed in one fixed order
        acc = dotr(Q, n, q, n, vx) + dotc(Jc, c, q, c, vyc) + dotc(Jg, g, q, g, vzn) - vd[q];
        if (!last)
          acc += dotr(eQ, n, q, n, vx) + dotr(M, n, q, m, vu) + dotc(A, n, q, n, vdn) + dotc(Jxc, ce, q, ce, vye) +
                 dotc(Jxg, ge, q, ge, vze);
      } else {
        acc = 0.0;
        if (pH)
          acc += dotr(Q, n, q, n, vx) + (last ? 0.0 : dotr(eQ, n, q, n, vx) + dotr(M, n, q, m, vu));
        if (pCT)
          acc += dotc(Jc, c, q, c, vyc) - vd[q] + (last ? 0.0 : dotc(A, n, q, n, vdn) + dotc(Jxc, ce, q, ce, vye));
        if (pGT)
          acc += dotc(Jg, g, q, g, vzn) + (last ? 0.0 : dotc(Jxg, ge, q, ge, vze));
      }
      *dst = yold + (acc + coef * vx[q]);
      continue;
    }
    q -= n;
    if (!last) {
      if (q < m) { // control rows
        double acc;
        if (all) {
          acc = dotc(M, n, q, n, vx) + dotr(R, m, q, m, vu) + dotc(B, n, q, n, vdn) + dotc(Juc, ce, q, ce, vye) +
                dotc(Jug, ge, q, ge, vze);
        } else {
          acc = 0.0;
          if (pH)
            acc += dotc(M, n, q, n, vx) + dotr(R, m, q, m, vu);
          if (pCT)
            acc += dotc(B, n, q, n, vdn) + dotc(Juc, ce, q, ce, vye);
          if (pGT)
            acc += dotc(Jug, ge, q, ge, vze);
        }
        *dst = yold + (acc + coef * vu[q]);
        continue;
      }
      q -= m;
      if (q < n) { // dynamics rows of node i + 1: A x + B u - x_{i+1} - r2 y
        const double acc = pC ? dotr(A, n, q, n, vx) + dotr(B, n, q, m, vu) - extra : 0.0;
        *dst = yold + (acc - coef * vdn[q]);
        continue;
      }
      q -= n;
    }
    if (q < c) {
      *dst = yold + ((pC ? dotr(Jc, c, q, n, vx) : 0.0) - coef * vyc[q]);
      continue;
    }
    q -= c;
    if (q < g) {
      *dst = yold + ((pG ? dotr(Jg, g, q, n, vx) : 0.0) - coef * vzn[q]);
      continue;
    }
    q -= g;
    if (q < ce) {
      *dst = yold + ((pC ? dotr(Jxc, ce, q, n, vx) + dotr(Juc, ce, q, m, vu) : 0.0) - coef * vye[q]);
      continue;
    }
    q -= ce;
    if (q < ge) {
      *dst = yold + ((pG ? dotr(Jxg, ge, q, n, vx) + dotr(Jug, ge, q, m, vu) : 0.0) - coef * vze[q]);
      continue;
    }
    q -= ge; // i == 0: the root's dynamics rows, -x_root - r2 y (helpers.cpp:1081-1085)
    *dst = yold + ((pC ? -vx[q] : 0.0) - coef * vd[q]);
  }
}

} // namespace kkt
} // namespace sipamd
