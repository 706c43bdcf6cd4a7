// chain_mrhs.hpp -- LQR::solve() for SEVERAL right-hand sides in one sweep (fp64, uniform chains).
//
// Replaces the multi-right-hand-side block of the reference's solve_stagewise_kkt_matrix
// (helpers.cpp:521-665: LQR::solve generalised from GEMV to GEMM over the columns of J_theta, reading
// LQR::Workspace directly) against the factor state a split factor launch (mode 1 of
// chain_factor_solve_qw16) left: S = F^{-1} per node in the spill, K in the gains, the LDL factors of
// the G matrices.  One launch carries `ncols` <= P columns through the backward affine sweep
// (lqr.cpp:738-796) and the rollout (lqr.cpp:821-870): every matrix operand of a stage is fetched
// once and applied to all columns, where the column-by-column path re-reads it per column
// (2 864 B per problem-stage and column at n = 12, m = 4).
//
// Mapping as the solve-only mode of chain_qw16.hpp: one problem per 16-lane DPP row, vectors
// DISTRIBUTED over the lanes (lane r holds element r), every matrix-vector product one
// broadcast-FMA block (dotv: acc += x@lane k * B[k]) on the lane's own column / row of the operand.
// The stage operands of the NEXT stage are requested before the columns of the current one are
// processed (registers), so the loop is not a chain of exposed HBM round trips.
//
// Per-column state that the rollout needs (g, h of the child node and k of the edge) goes to a
// column workspace: cws[problem][node i][col][g (N) | h (N) | k of edge i (M)].
//
// A launch of `ncols_launch` columns is a grid of (batch / 4) x ceil(ncols_launch / P) single-wave workgroups:
// wavefront (b, y) carries columns [y P, y P + P) of problems 4 b .. 4 b + 3.
#pragma once
#include <hip/hip_runtime.h>

#include "chain_qw16.hpp"

namespace sipamd {

template <int N, int M, bool WPACK, int P>
__global__ __launch_bounds__(64) void chain_solve_mrhs_qw16(
    const double *__restrict__ mats, const double *__restrict__ vecs_cols, double *__restrict__ sol_cols,
    const double *__restrict__ gains, const double *__restrict__ wsp, const double *__restrict__ gfac,
    double *__restrict__ cws, const int *__restrict__ status, const long batch, const int T, const int ncols_launch,
    const long col_stride /* scalars between two columns of vecs_cols / sol_cols */) {
  // blockIdx.y: the group of P columns this wavefront carries (the columns are independent: more wavefronts per
  // SIMD than one launch of batch / 4 single-wave workgroups gives, at the price of fetching the stage operands
  // once per group)
  const int col0 = (int)blockIdx.y * P;
  const int ncols = ncols_launch - col0 < P ? ncols_launch - col0 : P;
  static_assert(N >= 1 && N <= 16 && M >= 1 && M <= 16 && P >= 1 && P <= 16, "");
  using L = ChainLayout<N, M>;
  using C = StagedCfg<N, M, WPACK>;
  constexpr int STG = L::NODE + L::EDGE, VSTG = L::VNODE + L::VEDGE;
  constexpr int WSN = C::WSN;
  constexpr int CW = 2 * N + M; // g | h | k per (node, column)

  const int lane = threadIdx.x & 63, c = lane & 15, rr = lane >> 4;
  long p = (long)blockIdx.x * 4 + rr;
  const bool valid = p < batch;
  if (!valid)
    p = batch - 1;
  const bool isM = c < N;
  const int cm = isM ? c : N - 1, cu = c < M ? c : M - 1;
  // a problem whose factorization failed is skipped (LQR::solve after a failed factor is undefined in
  // the reference; the column-by-column path leaves such problems unspecified as well)
  const bool live = valid && status[p] == 0;

  const long mats_len = (long)(T + 1) * L::NODE + (long)T * L::EDGE;
  const long vecs_len = (long)(T + 1) * L::VNODE + (long)T * L::VEDGE;
  const double *pm = mats + p * mats_len;
  const double *pv = vecs_cols + p * vecs_len + col0 * col_stride;
  double *ps = sol_cols + p * vecs_len + col0 * col_stride;
  const double *pg = gains + p * ((long)T * L::GAIN);
  const double *pw = wsp + p * ((long)(T + 1) * WSN);
  const double *pf = gfac + p * ((long)T * (M * M + M));
  double *pc = cws + p * ((long)(T + 1) * ncols_launch * CW) + col0 * CW; // below: slot (i, col) at (i * ncols_launch + col) * CW

  int woff[N]; // S(row c, k) inside a spill slot (packed lower triangle or full, as the factor left it)
  sfor<0, N>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    if constexpr (WPACK) {
      const int lo = k < cm ? k : cm, hi = k < cm ? cm : k;
      woff[k] = lo * N - (lo * (lo - 1)) / 2 + (hi - lo);
    } else {
      woff[k] = cm * N + k;
    }
  });
  auto sum4 = [](const double (&a)[4]) { return (a[0] + a[1]) + (a[2] + a[3]); };

  // ---- backward affine sweep ------------------------------------------------------------------
  struct NodeOps { // node i: S row, delta, and per column c_i, q_i
    double Srow[N], dl, cvec[P], qvec[P];
  };
  struct EdgeOps { // edge i: columns of B, A; K column; the LDL factor of G; per column r_i
    double Bcol[N], Acol[N], Kc[M], Lf[M][M], rinv[M], rvec[P];
  };
  auto load_node = [&](const int i, NodeOps &o) {
    const double *slot = pw + (long)i * WSN;
    sfor<0, N>([&](auto kk) { o.Srow[decltype(kk)::value] = slot[woff[decltype(kk)::value]]; });
    o.dl = pm[(long)i * STG + N * N + cm];
    sfor<0, P>([&](auto cc) {
      constexpr int col = decltype(cc)::value;
      if (col < ncols) {
        o.qvec[col] = pv[col * col_stride + (long)i * VSTG + cm];
        o.cvec[col] = pv[col * col_stride + (long)i * VSTG + N + cm];
      }
    });
  };
  auto load_edge = [&](const int i, EdgeOps &o) {
    const double *em = pm + (long)i * STG + L::NODE;
    const double *gi = pg + (long)i * L::GAIN;
    const double *gf = pf + (long)i * (M * M + M);
    sfor<0, N>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      o.Bcol[k] = em[N * N + cu * N + k];
      o.Acol[k] = em[cm * N + k];
    });
    sfor<0, M>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      o.Kc[j] = gi[cm * M + j];
      o.rinv[j] = gf[M * M + j];
      sfor<0, j>([&](auto iv) { o.Lf[j][decltype(iv)::value] = gf[decltype(iv)::value * M + j]; });
    });
    sfor<0, P>([&](auto cc) {
      constexpr int col = decltype(cc)::value;
      if (col < ncols)
        o.rvec[col] = pv[col * col_stride + (long)i * VSTG + L::VNODE + cu];
    });
  };
  // t = c - delta o v; h = S D^{-1/2} t (kept for the rollout); returns W t = D^{-1/2}(I - S)D^{-1/2} t
  auto node_step = [&](const int i, const int col, const NodeOps &o, const double v_) {
    const double t_ = o.cvec[col] - o.dl * v_;
    const double sdi = rsqrt_nr(o.dl);
    const double phi = sdi * t_;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    dotv<N, true>(acc, phi, o.Srow);
    const double sphi = sum4(acc);
    if (live && isM)
      pc[((long)i * ncols_launch + col) * CW + N + c] = sphi;
    return sdi * (phi - sphi);
  };

  double v_[P], wt[P];
  {
    NodeOps term;
    load_node(T, term);
    sfor<0, P>([&](auto cc) {
      constexpr int col = decltype(cc)::value;
      if (col < ncols) {
        v_[col] = term.qvec[col]; // v_T = q_T
        wt[col] = node_step(T, col, term, v_[col]);
      }
    });
  }
  // One backward step over edge i with the operands in (ed, nd); the next stage's operands are
  // requested into (ed_next, nd_next) first, so that they travel while the columns are processed.
  // The loop below alternates two operand sets (no register copies between stages).
  auto backward_stage = [&](const int i, const EdgeOps &ed, const NodeOps &nd, EdgeOps &ed_next, NodeOps &nd_next) {
    if (i > 0) {
      load_edge(i - 1, ed_next);
      load_node(i - 1, nd_next);
    }
    sfor<0, P>([&](auto cc) {
      constexpr int col = decltype(cc)::value;
      if (col < ncols) {
        const double gd = v_[col] + wt[col]; // g = v_c + W t  (lqr.cpp:780-781)
        if (live && isM)
          pc[((long)(i + 1) * ncols_launch + col) * CW + c] = gd;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        dotv<N, true>(acc, gd, ed.Bcol);
        const double hd = ed.rvec[col] + sum4(acc); // h = r + B^T g  (:783-784), lane j < M holds h_j
        double hf[M], kf[M];
        sfor<0, M>([&](auto jj) { hf[decltype(jj)::value] = bcast<decltype(jj)::value>(hd); });
        sfor<0, M>([&](auto jj) { // k = -G^{-1} h  (:785-791), replicated over the lanes
          constexpr int j = decltype(jj)::value;
          double w = hf[j];
          sfor<0, j>([&](auto iv) { w = __builtin_fma(-ed.Lf[j][decltype(iv)::value], kf[decltype(iv)::value], w); });
          kf[j] = w * ed.rinv[j];
        });
        sfor_down<M - 1, -1>([&](auto jj) {
          constexpr int j = decltype(jj)::value;
          double a = 0.0;
          sfor<j + 1, M>([&](auto iv) { a = __builtin_fma(ed.Lf[decltype(iv)::value][j], kf[decltype(iv)::value], a); });
          kf[j] = __builtin_fma(-ed.rinv[j], a, kf[j]);
        });
        if (live && c == 0) {
          double *kd = pc + ((long)i * ncols_launch + col) * CW + 2 * N;
          sfor<0, M>([&](auto jj) { kd[decltype(jj)::value] = -kf[decltype(jj)::value]; });
        }
        double acc2[4] = {0.0, 0.0, 0.0, 0.0};
        dotv<N, true>(acc2, gd, ed.Acol);
        double kh = 0.0; // K^T h with K = -G^{-1} H (the stored gain)
        sfor<0, M>([&](auto jj) { kh = __builtin_fma(ed.Kc[decltype(jj)::value], hf[decltype(jj)::value], kh); });
        v_[col] = nd.qvec[col] + sum4(acc2) + kh; // v = q + A^T g + K^T h  (:793-794)
        wt[col] = node_step(i, col, nd, v_[col]);
      }
    });
  };
  NodeOps nd_a, nd_b;
  EdgeOps ed_a, ed_b;
  if (T > 0) {
    load_edge(T - 1, ed_a);
    load_node(T - 1, nd_a);
  }
  for (int i = T - 1; i >= 0; i -= 2) {
    backward_stage(i, ed_a, nd_a, ed_b, nd_b);
    if (i >= 1)
      backward_stage(i - 1, ed_b, nd_b, ed_a, nd_a);
  }
  sfor<0, P>([&](auto cc) { // root: g_0 = v_0 + W_0 (c_0 - delta_0 o v_0)  (lqr.cpp:798-819)
    constexpr int col = decltype(cc)::value;
    if (col < ncols && live && isM)
      pc[(long)col * CW + c] = v_[col] + wt[col];
  });
  // the rollout reads g / h / k written above by other lanes of this wave
  __syncthreads();

  // ---- forward rollout (lqr.cpp:821-870); lane r < N owns row r ----------------------------------
  struct FwdOps {
    double KT[N], Arow[N], Brow[M], Wc[N], dd, kk0[P], gg[P], hh[P];
  };
  auto load_fwd = [&](const int i, FwdOps &o) {
    const double *em = pm + (long)i * STG + L::NODE;
    const double *gi = pg + (long)i * L::GAIN;
    const double *wn = pw + (long)(i + 1) * WSN;
    sfor<0, N>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      o.KT[k] = gi[k * M + cu];
      o.Arow[k] = em[k * N + cm];
      o.Wc[k] = wn[woff[k]];
    });
    sfor<0, M>([&](auto jj) { o.Brow[decltype(jj)::value] = em[N * N + decltype(jj)::value * N + cm]; });
    o.dd = pm[(long)(i + 1) * STG + N * N + cm];
    sfor<0, P>([&](auto cc) {
      constexpr int col = decltype(cc)::value;
      if (col < ncols) {
        o.kk0[col] = pc[((long)i * ncols_launch + col) * CW + 2 * N + cu];
        o.gg[col] = pc[((long)(i + 1) * ncols_launch + col) * CW + cm];
        o.hh[col] = pc[((long)(i + 1) * ncols_launch + col) * CW + N + cm];
      }
    });
  };
  double x[P];
  {
    const double dd = pm[N * N + cm];
    const double sd0 = dd * rsqrt_nr(dd);
    sfor<0, P>([&](auto cc) { // root: x_0 = D^{1/2} h_0, y_0 = g_0
      constexpr int col = decltype(cc)::value;
      if (col < ncols) {
        const double gg = pc[(long)col * CW + cm], hh = pc[(long)col * CW + N + cm];
        x[col] = sd0 * hh;
        if (live && isM) {
          ps[col * col_stride + c] = x[col];
          ps[col * col_stride + N + c] = gg;
        }
      }
    });
  }
  auto forward_stage = [&](const int i, const FwdOps &fw, FwdOps &fw_next) {
    if (i + 1 < T)
      load_fwd(i + 1, fw_next);
    const double sdi = rsqrt_nr(fw.dd), sdv = fw.dd * sdi;
    sfor<0, P>([&](auto cc) {
      constexpr int col = decltype(cc)::value;
      if (col < ncols) {
        double acc[4] = {fw.kk0[col], 0.0, 0.0, 0.0};
        dotv<N, true>(acc, x[col], fw.KT);
        const double u = sum4(acc); // u = k + K x  (lqr.cpp:856-857); lanes < M
        double az[4] = {0.0, 0.0, 0.0, 0.0};
        dotv<N, true>(az, x[col], fw.Arow);
        dotv<M, true>(az, u, fw.Brow);
        const double zeta = sum4(az) * sdi; // D^{-1/2} (A x + B u)
        double as[4] = {0.0, 0.0, 0.0, 0.0};
        dotv<N, true>(as, zeta, fw.Wc);
        const double sz = sum4(as);
        x[col] = sdv * (sz + fw.hh[col]);                           // x_c = D^{1/2} (S zeta + h)
        const double y = __builtin_fma(sdi, zeta - sz, fw.gg[col]); // y_c = g_c + D^{-1/2} (zeta - S zeta)
        if (live) {
          double *si = ps + col * col_stride + (long)i * VSTG;
          if (c < M)
            si[2 * N + c] = u;
          if (isM) {
            si[VSTG + c] = x[col];
            si[VSTG + N + c] = y;
          }
        }
      }
    });
  };
  FwdOps fw_a, fw_b;
  if (T > 0)
    load_fwd(0, fw_a);
  for (int i = 0; i < T; i += 2) {
    forward_stage(i, fw_a, fw_b);
    if (i + 1 < T)
      forward_stage(i + 1, fw_b, fw_a);
  }
}

} // namespace sipamd
