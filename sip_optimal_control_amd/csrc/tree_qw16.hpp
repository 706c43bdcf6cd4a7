// tree_qw16.hpp -- fused factor + solve for TREES on the broadcast-FMA arithmetic of chain_qw16.hpp.
//
// The general engine (tree_generic.hpp) follows lqr.cpp with generic index arithmetic, one wavefront
// per problem: ~2 % of the HBM roofline.  This kernel gives any tree whose state dimensions are <= 15
// and control dimensions <= 8 the arithmetic of the chain kernels: one problem per 16-lane DPP row,
// lane c < N owns column c of every matrix, lane N carries the affine column, products are
// `v_fmac_f64_dpp row_newbcast` blocks.  Per-node dimensions are padded to ONE size class (N, M) IN
// REGISTERS as the blocks are loaded from the tree-native arenas (extra states / controls with Q = I,
// R = I, delta = 1 on their diagonal and zeros elsewhere decouple exactly: x = y = u = 0 there), so the
// arithmetic is that of a uniform tree and no padded copy of the inputs or outputs ever exists.
//
// Traversal as the reference (lqr.cpp:645-731, 735-871): nodes in postorder, the child edges of a node
// in CSR order (so a failing factorization reports the status of the first failing node / edge, G
// before delta before F at a node); rollout in preorder.  What a chain keeps in registers from one
// stage to the next, a tree fetches from the per-node spill when the PARENT is processed:
//     spill[node] = S = F^{-1} (N x N) | g | h | t = c - delta o v | v
// W of a child is rebuilt from its S (W = D^{-1/2}(I - S)D^{-1/2}, lqr.cpp:521-528), the affine
// column of [F | g] = W [A | t] + [0 | v] starts from the child's (t, v), and V of the parent
// accumulates A^T F + K^T H over its child edges (lqr.cpp:715-719).  The rollout reads the parent's x
// back from the solution arena.  Sibling subtrees are processed one after the other by the same
// 16-lane row: with a batch that fills the machine (4 problems per wavefront) sibling parallelism
// inside a problem has nothing left to fill.
//
// Inputs, outputs and (optionally) K, k: the tree-native arenas of include/sip_lqr_amd.h.  Internal
// scratch (per problem, doubles, padded to the size class):
//   gains: edge e: K (M x N) | k (M)        spill: node i: S | g | h | t | v
#pragma once
#include <hip/hip_runtime.h>

#include "chain_qw16.hpp"

namespace sipamd {

struct TreeTopo { // device arrays of the compiled traversal (GenericPlan)
  int E, Nn, root;
  const int *parents, *children, *child_offsets, *child_edges, *preorder, *postorder;
};

// Tree-native arenas of include/sip_lqr_amd.h (per-node / per-edge dimensions, offsets in doubles
// from the start of one problem) as the kernel reads and writes them directly.
struct TreeNative {
  const int *sd, *cd;                                      // state dims per node, control dims per edge
  const long *oQ, *oq, *oc, *od, *oA, *oB, *oM, *oR, *orr; // input arena
  const long *ox, *oy, *ou;                                // output arena
  const long *oK, *ok;                                     // work arena: K, k of every edge
  long in_len, out_len, ws_len;
};

template <int N, int M>
struct TreeLayout {
  static constexpr int GAIN = M * N + M;   // K (M x N, padded) | k
  static constexpr int WS = N * N + 4 * N; // S | g | h | t | v
};

template <int N, int M>
__global__ __launch_bounds__(64) void tree_factor_solve_qw16(
    const TreeTopo tp, const TreeNative tn, const double *__restrict__ in_all, double *__restrict__ out_all,
    double *__restrict__ work_all /* may be null: K, k not wanted */, double *__restrict__ gains,
    double *__restrict__ wsp, int *__restrict__ status, const long batch) {
  static_assert(N >= 1 && N <= 15, "one lane of the row carries the affine column");
  static_assert(M >= 1 && M <= 16, "");
  using L = TreeLayout<N, M>;
  const int lane = threadIdx.x & 63, c = lane & 15, rr = lane >> 4;
  long p = (long)blockIdx.x * 4 + rr;
  const bool valid = p < batch;
  if (!valid)
    p = batch - 1;
  const bool isM = c < N, isV = c == N;
  const int Nn = tp.Nn, E_ = tp.E;
  const double *in = in_all + p * tn.in_len;
  double *out = out_all + p * tn.out_len;
  double *work = work_all != nullptr ? work_all + p * tn.ws_len : nullptr;
  double *pg = gains + p * ((long)E_ * L::GAIN); // padded gains: what the rollout reads back
  double *pw = wsp + p * ((long)Nn * L::WS);

  double E[N];
  sfor<0, N>([&](auto ii) { E[decltype(ii)::value] = (c == decltype(ii)::value) ? 1.0 : 0.0; });

  // Column `col` (clamped into the block) of a column-major rows x cols block, padded to N rows:
  // dst[r] = on && r < rows ? blk[r + rows * col] : fill[r].  `rows`, `cols` are wave-uniform.
  auto load_col = [&](double (&dst)[N], const double *blk, const int rows, const int cols, const int col,
                      const bool on, const double (&fill)[N]) {
    const int cc = col < cols ? col : (cols > 0 ? cols - 1 : 0);
    const bool use = on && col < cols;
    sfor<0, N>([&](auto ii) {
      constexpr int r = decltype(ii)::value;
      double v = fill[r];
      if (r < rows && cols > 0) { // wave-uniform
        const double ld = blk[r + (long)rows * cc];
        v = use ? ld : v;
      }
      dst[r] = v;
    });
  };
  double ZERO[N];
  sfor<0, N>([&](auto ii) { ZERO[decltype(ii)::value] = 0.0; });

  int stat = 0;
  double W[N], V[N], tv[N];

  // ---- backward: nodes in postorder (lqr.cpp:651) ------------------------------------------------
  for (int idx = 0; idx < Nn; ++idx) {
    const int j = tp.postorder[idx];
    const int n = tn.sd[j];
    const double *Qj = in + tn.oQ[j], *qj = in + tn.oq[j], *cj = in + tn.oc[j], *dj = in + tn.od[j];
    // [V | v] = [Q | q]  (lqr.cpp:658, :744); identity columns on the padding lanes
    if (isV) {
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        V[r] = r < n ? qj[r] : 0.0;
      });
    } else {
      load_col(V, Qj, n, n, c, isM, E);
      if (!isM)
        sfor<0, N>([&](auto ii) { V[decltype(ii)::value] = 0.0; });
    }
    for (int ci = tp.child_offsets[j]; ci < tp.child_offsets[j + 1]; ++ci) { // lqr.cpp:660
      const int e = tp.child_edges[ci], ch = tp.children[e];
      const int nc = tn.sd[ch], m = tn.cd[e];
      const double *Ae = in + tn.oA[e], *Be = in + tn.oB[e], *Me = in + tn.oM[e], *Re = in + tn.oR[e];
      const double *re = in + tn.orr[e], *dch = in + tn.od[ch];
      double *slot = pw + (long)ch * L::WS;
      const int cmN = isM ? c : N - 1;
      // W of the child from its spilled S (lqr.cpp:521-528)
      {
        const double dlc = (c < nc) ? dch[c < nc ? c : 0] : 1.0;
        const double sdi = rsqrt_nr(dlc);
        double Sc[N], scale[N];
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          Sc[r] = slot[cmN * N + r];
          scale[r] = 0.0;
        });
        spread<N, false, true>(scale, sdi, sdi); // sdi_r sdi_c
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          W[r] = (E[r] - Sc[r]) * scale[r];
        });
      }
      double F[N], Aaug[N], Bcol[N], Hc[M], G[M], rinvG[M], H[M], K[M];
      // column c of R (identity on the padding), column c of M^T = row c of M; vector lane: r
      sfor<0, M>([&](auto jj) {
        constexpr int q = decltype(jj)::value;
        double gq = (q == c) ? 1.0 : 0.0, hq = 0.0;
        if (q < m) { // wave-uniform
          const double rl = Re[q + (long)m * (c < m ? c : 0)];
          gq = c < m ? rl : gq;
          const double rv = re[q];
          const double ml = n > 0 ? Me[(c < n ? c : 0) + (long)n * q] : 0.0;
          hq = isV ? rv : (c < n ? ml : 0.0);
        }
        G[q] = gq, H[q] = hq;
      });
      // columns of A (nc x n) and B (nc x m); vector lane of Aaug / F: t and v of the child
      load_col(Aaug, Ae, nc, n, c, isM, ZERO);
      load_col(Bcol, Be, nc, m, c, true, ZERO);
      if (isV)
        sfor<0, N>([&](auto kk) { Aaug[decltype(kk)::value] = slot[N * N + 2 * N + decltype(kk)::value]; });
      sfor<0, N>([&](auto kk) { F[decltype(kk)::value] = isV ? slot[N * N + 3 * N + decltype(kk)::value] : 0.0; });
      rank1x<N, N, true>(F, W, Aaug); // [F | g] = W [A | t] + [0 | v]  (lqr.cpp:703, :780-781)
      if (valid && isV)
        sfor<0, N>([&](auto ii) { slot[N * N + decltype(ii)::value] = F[decltype(ii)::value]; }); // g of the child
      // H_child = B^T W (lqr.cpp:692); G = R + H_child B (:693-694)
      sfor<0, M>([&](auto jj) { Hc[decltype(jj)::value] = 0.0; });
      spreadx<M, N, false>(Hc, Bcol, W);
      rank1x<M, N, true>(G, Hc, Bcol);
      const bool gfail = chol_ldl_dpp<M>(G, rinvG, c); // lqr.cpp:696-701
      if (stat == 0 && gfail)
        stat = 3; // G_FACTORIZATION_FAILURE
      spreadx<M, N, false>(H, Bcol, F); // [H | h] = [M^T | r] + B^T [F | g]  (:704-705, :783-784)
      sfor<0, M>([&](auto jj) { K[decltype(jj)::value] = H[decltype(jj)::value]; });
      ldl_solve_dpp<M>(G, rinvG, K); // [K | k] = -G^{-1} [H | h]  (:707-713, :785-791)
      sfor<0, M>([&](auto jj) { K[decltype(jj)::value] = -K[decltype(jj)::value]; });
      if (valid && c <= N) {
        double *gi = pg + (long)e * L::GAIN + c * M;
        sfor<0, M>([&](auto jj) { gi[decltype(jj)::value] = K[decltype(jj)::value]; });
        if (work != nullptr) { // LQR::Workspace::K (m x n, column-major) and k of the caller's work arena
          double *dst = isV ? work + tn.ok[e] : work + tn.oK[e] + (long)m * c;
          if (isV || c < n)
            sfor<0, M>([&](auto jj) {
              constexpr int q = decltype(jj)::value;
              if (q < m)
                dst[q] = K[q];
            });
        }
      }
      // [V | v] += A^T [F | g] + K^T [H | h]  (lqr.cpp:715-719, :793-794)
      spreadx<N, N, false>(V, Aaug, F);
      spreadx<N, M, true>(V, K, H);
      asm volatile("" ::: "memory");
    }
    // the node itself: statuses, F / S, the affine terms the parent step needs (lqr.cpp:722-727)
    double *mine = pw + (long)j * L::WS;
    const double dl = (c < n) ? dj[c < n ? c : 0] : 1.0;
    {
      const unsigned long long bad = __ballot(c < n && dl <= 0.0);
      if (stat == 0 && ((bad >> (lane & 48)) & 0xffffull) != 0)
        stat = 1; // INVALID_DELTA
    }
    sfor<0, N>([&](auto ii) {
      constexpr int r = decltype(ii)::value;
      double cv = 0.0, dv = 0.0;
      if (r < n) { // wave-uniform
        const double cl = cj[r], dlr = dj[r];
        cv = isV ? cl : 0.0, dv = isV ? dlr : 0.0;
      }
      tv[r] = cv - dv * V[r]; // c - delta o v on the vector lane, zeros elsewhere
    });
    if (valid && isV)
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        mine[N * N + 2 * N + r] = tv[r];
        mine[N * N + 3 * N + r] = V[r];
      });
    double X[N];
    const bool ffail = node_factor<N>(V, dl, c, E, tv, W, X);
    if (stat == 0 && ffail)
      stat = 2; // F_FACTORIZATION_FAILURE
    if (valid && isV) // h = S D^{-1/2} (c - delta o v)
      sfor<0, N>([&](auto ii) { mine[N * N + N + decltype(ii)::value] = X[decltype(ii)::value]; });
    if (valid && isM)
      sfor<0, N>([&](auto ii) { mine[c * N + decltype(ii)::value] = X[decltype(ii)::value]; });
    asm volatile("" ::: "memory");
  }
  // root (last in postorder): g = v + W (c - delta o v)  (lqr.cpp:798-819)
  {
    double F[N];
    sfor<0, N>([&](auto ii) { F[decltype(ii)::value] = isV ? V[decltype(ii)::value] : 0.0; });
    rank1x<N, N, true>(F, W, tv);
    if (valid && isV) {
      double *gn = pw + (long)tp.root * L::WS + N * N;
      sfor<0, N>([&](auto ii) { gn[decltype(ii)::value] = F[decltype(ii)::value]; });
    }
  }
  if (valid && c == 0)
    status[p] = stat;
  // the rollout reads S / g / h / K / k written above by other lanes of this wave
  __syncthreads();

  // ---- forward rollout in preorder (lqr.cpp:821-870); lane r < N owns row r -----------------------
  auto sum4 = [](const double (&a)[4]) { return (a[0] + a[1]) + (a[2] + a[3]); };
  const int cmN = isM ? c : N - 1, cuM = c < M ? c : M - 1;
  for (int idx = 0; idx < Nn; ++idx) {
    const int j = tp.preorder[idx];
    const int n = tn.sd[j];
    double x;
    if (idx == 0) { // x_root = D^{1/2} h_root, y_root = g_root
      const double *slot = pw + (long)j * L::WS;
      const double dd = (c < n) ? in[tn.od[j] + (c < n ? c : 0)] : 1.0;
      x = (dd * rsqrt_nr(dd)) * slot[N * N + N + cmN];
      if (valid && c < n) {
        out[tn.ox[j] + c] = x;
        out[tn.oy[j] + c] = slot[N * N + cmN];
      }
      if (c >= n)
        x = 0.0;
    } else {
      x = (c < n) ? out[tn.ox[j] + (c < n ? c : 0)] : 0.0; // written by this lane when the parent was rolled out
    }
    for (int ci = tp.child_offsets[j]; ci < tp.child_offsets[j + 1]; ++ci) {
      const int e = tp.child_edges[ci], ch = tp.children[e];
      const int nc = tn.sd[ch], m = tn.cd[e];
      const double *Ae = in + tn.oA[e], *Be = in + tn.oB[e];
      const double *gi = pg + (long)e * L::GAIN;
      const double *slot = pw + (long)ch * L::WS;
      double KT[N], Arow[N], Brow[M], Wc[N];
      sfor<0, N>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        KT[k] = gi[k * M + cuM];  // padded gains: zeros on the padding
        Wc[k] = slot[cmN * N + k]; // S symmetric: row c = column c
        double a = 0.0;
        if (k < n && nc > 0) { // wave-uniform: row c of A (nc x n)
          const double al = Ae[(c < nc ? c : 0) + (long)nc * k];
          a = c < nc ? al : 0.0;
        }
        Arow[k] = a;
      });
      sfor<0, M>([&](auto jj) {
        constexpr int q = decltype(jj)::value;
        double b = 0.0;
        if (q < m && nc > 0) {
          const double bl = Be[(c < nc ? c : 0) + (long)nc * q];
          b = c < nc ? bl : 0.0;
        }
        Brow[q] = b;
      });
      const double kk0 = gi[N * M + cuM], gg = slot[N * N + cmN], hh = slot[N * N + N + cmN];
      const double dd = (c < nc) ? in[tn.od[ch] + (c < nc ? c : 0)] : 1.0;
      const double sdi = rsqrt_nr(dd), sdv = dd * sdi;
      double acc[4] = {kk0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, x, KT);
      const double u = sum4(acc); // u = k + K x  (lqr.cpp:856-857)
      double az[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(az, x, Arow);
      dotv<M, true>(az, u, Brow);
      const double zeta = sum4(az) * sdi; // D^{-1/2} (A x + B u)
      double as[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(as, zeta, Wc);
      const double sz = sum4(as);
      const double xc = sdv * (sz + hh);                   // x_c = D^{1/2} (S zeta + h)
      const double yc = __builtin_fma(sdi, zeta - sz, gg); // y_c = g_c + D^{-1/2} (zeta - S zeta)
      if (valid) {
        if (c < m)
          out[tn.ou[e] + c] = u;
        if (c < nc) {
          out[tn.ox[ch] + c] = xc;
          out[tn.oy[ch] + c] = yc;
        }
      }
    }
  }
}

} // namespace sipamd
