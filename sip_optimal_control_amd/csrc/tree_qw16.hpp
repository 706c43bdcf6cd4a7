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
// Traversal as the reference (lqr.cpp:645-731, 735-871), flattened on the host into one record per
// step (TreeStep): nodes in postorder, the child edges of a node
// in CSR order (so a failing factorization reports the status of the first failing node / edge, G
// before delta before F at a node); rollout in preorder.  What a chain keeps in registers from one
// stage to the next, a tree fetches from the per-node spill when the PARENT is processed:
//     spill[node] = S = F^{-1} (N x N) | g | h | t = c - delta o v | v
// W of a child is rebuilt from its S (W = D^{-1/2}(I - S)D^{-1/2}, lqr.cpp:521-528), the affine
// column of [F | g] = W [A | t] + [0 | v] starts from the child's (t, v), and V of the parent
// accumulates A^T F + K^T H over its child edges (lqr.cpp:715-719).  The rollout reads the parent's x
// back from the solution arena.  The FIRST child of a node is always the node finished immediately
// before it (postorder = reversed preorder), so its W, t, v are taken from registers instead (a chain
// never reads its spill in the backward sweep), and likewise x of the parent in the rollout.  Sibling
// subtrees are processed one after the other by the same
// 16-lane row: with a batch that fills the machine (4 problems per wavefront) sibling parallelism
// inside a problem has nothing left to fill.
//
// Inputs, outputs and (optionally) the factor state of LQR::Workspace (lqr.hpp:109-135: W, K, G_factor, k per
// edge; V, F_factor, sqrt_delta, sqrt_delta_inv, v per node -- what helpers.cpp:521-665 reads): the tree-native
// arenas of include/sip_lqr_amd.h.  Internal
// scratch (per problem, doubles, padded to the size class):
//   gains: edge e: K (M x N) | k (M)        spill: node i: S | g | h | t | v
#pragma once
#include <hip/hip_runtime.h>

#include "chain_qw16.hpp"

namespace sipamd {

// One step of the flattened traversal (built on the host from the compiled topology, one table per
// plan): everything a step needs is ONE record read by scalar loads, instead of a chain of dependent
// table look-ups (postorder -> child_offsets -> child_edges -> children -> dims -> offsets) in front
// of every step.  Offsets are in doubles from the start of one problem inside its tree-native arena
// (include/sip_lqr_amd.h).
//   backward: for node in postorder { EDGE step per child edge, in CSR order; then the NODE step }
//   forward : one step per edge, parents in preorder, child edges in CSR order
struct TreeStep {
  int kind;   // backward: 0 = edge, 1 = node
  int node;   // edge steps: the parent; node steps: the node
  int edge, child;
  int n, nc, m; // dim(node), dim(child), control dim of the edge
  int flags;    // TS_*
  long oQ, oq, oc, od;             // input arena, of `node`
  long oA, oB, oM, oR, orr, odc;   // input arena, of the edge; delta of the child
  long oK, ok;                     // work arena: K, k of the edge
  long oW, oG;                     // work arena: W, G_factor of the edge (LQR::Workspace, lqr.hpp:109-135)
  long oV, oF, osd, osdi, ov;      // work arena: V, F_factor, sqrt_delta, sqrt_delta_inv, v of `node`
  long ou, oxc, oyc, oxp;          // output arena: u of the edge, x / y of the child, x of the parent
};
enum {
  TS_LOAD_V = 1,    // backward: [V | v] = [Q | q] of `node` first (first child edge of a node / a leaf's node step)
  TS_CHILD_LIVE = 2 // backward edge: the child was finished by the previous step: its W, t, v are still in registers
                    // forward edge: x of the parent is the x the previous step produced (still in registers)
};

struct TreeSchedule { // device tables + sizes
  const TreeStep *backward, *forward;
  int n_backward, n_forward; // Nn + E, E
  int Nn, E, root, root_n;
  long root_od, root_ox, root_oy;
  long in_len, out_len, ws_len;
};

#ifdef SIP_TREE_STAMPS // diagnostic build (tools/tree_ab_build.sh ... -DSIP_TREE_STAMPS): cycles per segment, summed over wavefronts
__device__ unsigned long long g_tree_seg[8];
#define TREE_SEG(k)                                                                                  \
  do {                                                                                               \
    unsigned long long now_;                                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                     \
    seg[k] += now_ - seg_last;                                                                       \
    seg_last = now_;                                                                                 \
  } while (0)
#define TREE_SEG_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define TREE_SEG(k)                                                                                  \
  do {                                                                                               \
  } while (0)
#define TREE_SEG_DRAIN()                                                                             \
  do {                                                                                               \
  } while (0)
#endif

template <int N, int M>
struct TreeLayout {
  static constexpr int GAIN = M * N + M;   // K (M x N, padded) | k
  static constexpr int WS = N * N + 4 * N; // S | g | h | t | v
};

// EXPORT: also write W, G_factor (edges) and V, F_factor, sqrt_delta, sqrt_delta_inv, v (nodes) of LQR::Workspace
// into the work arena (which must then be given): a second instantiation, so that the kernel behind
// sip_lqr_tree_factor_solve keeps its registers.
template <int N, int M, bool EXPORT = false>
__global__ __launch_bounds__(64) void tree_factor_solve_qw16(
    const TreeSchedule ts, const double *__restrict__ in_all, double *__restrict__ out_all,
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
  const int cmN = isM ? c : N - 1, cuM = c < M ? c : M - 1;
  const double *in = in_all + p * ts.in_len;
  double *out = out_all + p * ts.out_len;
  double *work = work_all != nullptr ? work_all + p * ts.ws_len : nullptr;
  double *pg = gains + p * ((long)ts.E * L::GAIN); // padded gains: what the rollout reads back
  double *pw = wsp + p * ((long)ts.Nn * L::WS);

  double E[N], ZERO[N];
  sfor<0, N>([&](auto ii) {
    E[decltype(ii)::value] = (c == decltype(ii)::value) ? 1.0 : 0.0;
    ZERO[decltype(ii)::value] = 0.0;
  });

  auto E_of = [&](const int k) { return c == k ? 1.0 : 0.0; };
  // Column `col` of a column-major rows x cols block, padded to N rows:
  // dst[r] = on && col < cols && r < rows ? blk[r + rows * col] : fill[r].  Every load is issued
  // unconditionally from a clamped (always readable) address and the padding is selected afterwards:
  // no branches, the loads of a block travel together.
  auto load_col = [&](double (&dst)[N], const double *blk, const int rows, const int cols, const int col,
                      const bool on, const double (&fill)[N]) {
    const bool any = rows > 0 && cols > 0;
    const double *b = any ? blk : in; // an empty block: read the arena's first scalar instead (never used)
    const int cc = any ? (col < cols ? col : cols - 1) : 0;
    const bool use = on && any && col < cols;
    const double *colp = b + (long)(any ? rows : 0) * cc;
    sfor<0, N>([&](auto ii) {
      constexpr int r = decltype(ii)::value;
      // (both arms of every selection below are plain locals and the conditions are combined with `&`: the
      // front end then emits a select at once.  With an array element or a short-circuit `&&` in the ternary it
      // emits a branch diamond first, the optimizer sinks the load into the taken arm, and the "unconditional"
      // load comes out as an exec-masked region of its own -- 42 of them per backward step before this.)
      const double ld = colp[(any & (r < rows)) ? r : 0];
      const double fl = fill[r];
      const bool take = use & (r < rows);
      dst[r] = take ? ld : fl;
    });
  };
  // element `idx` of a vector of `len` entries, `fill` past its end or when !on
  auto load_elem = [&](const double *vec, const int len, const int idx, const bool on, const double fill) {
    const double ld = (len > 0 ? vec : in)[((len > 0) & (idx < len)) ? idx : 0];
    const bool take = on & (idx < len);
    return take ? ld : fill;
  };

  // L2 prefetch of a span of `len` scalars at `span` (this row's problem): every lane pulls one
  // dword of a different 64-byte line into a 256-byte LDS sink by LDS-DMA -- no destination
  // registers, nothing ever waits for it -- so the demand loads of the NEXT step find their lines in
  // the XCD's L2 instead of paying an HBM round trip at the top of every step.  LINES16: how many
  // groups of 16 lines (1 KiB per problem) the span may need.
  // (Not in the largest class: (15, 8) fills the register file -- 256 VGPRs + 240 AGPRs -- and the
  // prefetch's few extra live values would push it into scratch.)
  __shared__ unsigned prefetch_sink[64];
#ifdef SIP_TREE_NO_PREFETCH
  constexpr bool kPrefetch = false;
#else
  // (nor in the workspace-export instantiation of the 15-state classes, which spills: scratch traffic and LDS-DMA
  // share vmcnt, tools/check_dpp_hazards.py)
  constexpr bool kPrefetch = N * (N + 2 * M) <= 400 && !(EXPORT && N >= 15);
#endif
  auto prefetch = [&](const double *span, const long len, auto groups) {
    if constexpr (kPrefetch) {
    const long last = len > 0 ? (len - 1) * 8 : 0; // byte offset of the span's last scalar
    const char *b = (const char *)(len > 0 ? span : in);
    sfor<0, decltype(groups)::value>([&](auto gg) {
      long off = (long)(decltype(gg)::value * 16 + c) * 64;
      off = off < last ? off : last;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + off),
                                       (__attribute__((address_space(3))) void *)prefetch_sink, 4, 0, 0);
    });
    }
  };
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, (N * (N + 2 * M) + M * M + M + 127) / 128>; // an edge block A | B | M | R | r
  using GW = std::integral_constant<int, (L::WS + 127) / 128>;                       // a spill slot

  int stat = 0;
  double W[N], V[N], tv[N]; // W / tv: of the node finished last; V: accumulator of the current parent
  double Vc[N];             // [V | v] of the node finished last (its vector lane: v), for TS_CHILD_LIVE

  // ---- backward (lqr.cpp:645-731 and :738-796) ------------------------------------------------------
  // Input blocks of a step (read-only data), requested together at the top of the step.
  struct Pre {
    double a[N], b[N], g[M], h[M], v[N], dl;
  };
  auto fetch = [&](const TreeStep &st, Pre &o) {
    const int n = st.n, nc = st.nc, m = st.m;
    if (st.flags & TS_LOAD_V) { // [Q | q] of `node`; identity columns on the padding lanes
      const double *src = isV ? in + st.oq : in + st.oQ;
      load_col(o.v, src, n, isV ? 1 : n, isV ? 0 : c, isM || isV, E);
    }
    if (st.kind == 0) {
      // column c of R (identity on the padding), column c of M^T = row c of M; vector lane: r
      const bool anyR = m > 0, anyM = m > 0 && n > 0;
      const double *Rc = (anyR ? in + st.oR : in) + (long)(anyR ? m : 0) * (c < m ? c : 0);
      const double *Mr = (anyM ? in + st.oM : in) + (anyM && c < n ? c : 0);
      const double *rv = anyR ? in + st.orr : in;
      sfor<0, M>([&](auto jj) {
        constexpr int q = decltype(jj)::value;
        const int qq = (anyR & (q < m)) ? q : 0;
        const double rl = Rc[qq], ml = Mr[(long)(anyM ? n : 0) * qq], rr_ = rv[qq];
        const double unit = (q == c) ? 1.0 : 0.0, mz = (c < n) ? ml : 0.0, hv = isV ? rr_ : mz;
        const bool inR = (q < m) & (c < m);
        o.g[q] = inR ? rl : unit;
        o.h[q] = (q < m) ? hv : 0.0;
      });
      load_col(o.a, in + st.oA, nc, n, c, isM, ZERO); // column c of A (nc x n)
      load_col(o.b, in + st.oB, nc, m, c, true, ZERO); // column c of B (nc x m)
      o.dl = load_elem(in + st.odc, nc, c, true, 1.0); // delta of the child
    } else {
      load_col(o.a, in + st.oc, n, 1, 0, isV, ZERO); // c on the vector lane
      load_col(o.b, in + st.od, n, 1, 0, isV, ZERO); // delta on the vector lane
      o.dl = load_elem(in + st.od, n, c, true, 1.0);
    }
  };
  auto backward_step = [&](const TreeStep &st, const Pre &pre) {
    const int n = st.n;
    if (st.flags & TS_LOAD_V) // [V | v] = [Q | q]  (lqr.cpp:658, :744)
      sfor<0, N>([&](auto ii) { V[decltype(ii)::value] = pre.v[decltype(ii)::value]; });
    if (st.kind == 0) { // one child edge of `node` (lqr.cpp:660-720)
      const int e = st.edge, m = st.m;
      double *slot = pw + (long)st.child * L::WS;
      double F[N], Aaug[N], Hc[M], G[M], rinvG[M], H[M], K[M];
      if (st.flags & TS_CHILD_LIVE) {
        // the child is the node the previous step finished: W, t = tv and v = Vc are in registers
        sfor<0, N>([&](auto kk) {
          constexpr int k = decltype(kk)::value;
          Aaug[k] = isV ? tv[k] : pre.a[k];
          F[k] = isV ? Vc[k] : 0.0;
        });
      } else { // W of the child from its spilled S (lqr.cpp:521-528), t and v from its slot
        const double sdi = rsqrt_nr(pre.dl);
        double Sc[N], scale[N];
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          Sc[r] = slot[cmN * N + r];
          scale[r] = 0.0;
          const double tl = slot[N * N + 2 * N + r], vl = slot[N * N + 3 * N + r], al = pre.a[r]; // (locals: see load_col)
          Aaug[r] = isV ? tl : al;
          F[r] = isV ? vl : 0.0;
        });
        spread<N, false, true>(scale, sdi, sdi); // sdi_r sdi_c
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          W[r] = (E[r] - Sc[r]) * scale[r];
        });
      }
      sfor<0, M>([&](auto jj) {
        G[decltype(jj)::value] = pre.g[decltype(jj)::value];
        H[decltype(jj)::value] = pre.h[decltype(jj)::value];
      });
      if (EXPORT && valid && c < st.nc) { // LQR::Workspace::W of the edge (nc x nc, column-major)
        double *dst = work + st.oW + (long)st.nc * c;
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          if (r < st.nc)
            dst[r] = W[r];
        });
      }
      rank1x<N, N, true>(F, W, Aaug); // [F | g] = W [A | t] + [0 | v]  (lqr.cpp:703, :780-781)
      if (valid && isV)
        sfor<0, N>([&](auto ii) { slot[N * N + decltype(ii)::value] = F[decltype(ii)::value]; }); // g of the child
      // H_child = B^T W (lqr.cpp:692); G = R + H_child B (:693-694)
      sfor<0, M>([&](auto jj) { Hc[decltype(jj)::value] = 0.0; });
      spreadx<M, N, false>(Hc, pre.b, W);
      rank1x<M, N, true>(G, Hc, pre.b);
      double G0[M]; // export only: the entries above the diagonal keep their pre-factor values (Eigen's in-place LLT)
      if constexpr (EXPORT)
        sfor<0, M>([&](auto jj) { G0[decltype(jj)::value] = G[decltype(jj)::value]; });
      const bool gfail = chol_ldl_dpp<M>(G, rinvG, c); // lqr.cpp:696-701
      if (stat == 0 && gfail)
        stat = 3; // G_FACTORIZATION_FAILURE
      if (EXPORT && valid && c < m) {
        // LQR::Workspace::G_factor: lower triangle L of G = L L^T; the kernel holds Lt(i, c) = L(i, c) L(c, c)
        double dc = 0.0;
        sfor<0, M>([&](auto jj) { dc = __builtin_fma(E_of(decltype(jj)::value), G[decltype(jj)::value], dc); }); // pivot d_c
        const double li = rsqrt_nr(dc); // 1 / L(c, c)
        double *dst = work + st.oG + (long)m * c;
        sfor<0, M>([&](auto jj) {
          constexpr int q = decltype(jj)::value;
          if (q < m)
            dst[q] = q >= c ? G[q] * li : G0[q];
        });
      }
      spreadx<M, N, false>(H, pre.b, F); // [H | h] = [M^T | r] + B^T [F | g]  (:704-705, :783-784)
      sfor<0, M>([&](auto jj) { K[decltype(jj)::value] = H[decltype(jj)::value]; });
      ldl_solve_dpp<M>(G, rinvG, K); // [K | k] = -G^{-1} [H | h]  (:707-713, :785-791)
      sfor<0, M>([&](auto jj) { K[decltype(jj)::value] = -K[decltype(jj)::value]; });
      if (valid && c <= N) {
        double *gi = pg + (long)e * L::GAIN + c * M;
        sfor<0, M>([&](auto jj) { gi[decltype(jj)::value] = K[decltype(jj)::value]; });
        if (work != nullptr) { // LQR::Workspace::K (m x n, column-major) and k of the caller's work arena
          double *dst = isV ? work + st.ok : work + st.oK + (long)m * c;
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
    } else { // the node itself: statuses, F / S, the affine terms the parent step needs (lqr.cpp:722-727)
      double *mine = pw + (long)st.node * L::WS;
      const double dl = pre.dl;
      {
        const unsigned long long bad = __ballot(c < n && dl <= 0.0);
        if (stat == 0 && ((bad >> (lane & 48)) & 0xffffull) != 0)
          stat = 1; // INVALID_DELTA
      }
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        tv[r] = pre.a[r] - pre.b[r] * V[r]; // c - delta o v on the vector lane, zeros elsewhere
        Vc[r] = V[r];
      });
      if (valid && isV)
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          mine[N * N + 2 * N + r] = tv[r];
          mine[N * N + 3 * N + r] = V[r];
        });
      double X[N];
      bool ffail;
      if constexpr (EXPORT) { // the same factorization, with the factor state of LQR::Workspace written out
        double Lt[N];
        ffail = node_factor<N, true>(V, dl, c, E, tv, W, X, Lt);
        if (valid && c < n) {
          const double sdi = rsqrt_nr(dl), sd = dl * sdi;
          work[st.osd + c] = sd, work[st.osdi + c] = sdi;
          double dc = 0.0, sdr[N];
          sfor<0, N>([&](auto ii) {
            dc = __builtin_fma(E[decltype(ii)::value], Lt[decltype(ii)::value], dc); // pivot d_c = L(c, c)^2
            sdr[decltype(ii)::value] = 0.0;
          });
          spread<N, false, true>(sdr, sd, sd); // sd_r sd_c
          const double li = rsqrt_nr(dc);
          double *dv = work + st.oV + (long)n * c, *df = work + st.oF + (long)n * c;
          sfor<0, N>([&](auto ii) {
            constexpr int r = decltype(ii)::value;
            if (r < n) {
              dv[r] = V[r];
              // F_factor: L below and on the diagonal, the pre-factor I + D^1/2 V D^1/2 above (lqr.cpp:497-505)
              df[r] = r >= c ? Lt[r] * li : __builtin_fma(sdr[r], V[r], E[r]);
            }
          });
        }
        if (valid && isV)
          sfor<0, N>([&](auto ii) {
            constexpr int r = decltype(ii)::value;
            if (r < n)
              work[st.ov + r] = V[r]; // the affine column: v of the node
          });
      } else {
        ffail = node_factor<N>(V, dl, c, E, tv, W, X);
      }
      if (stat == 0 && ffail)
        stat = 2; // F_FACTORIZATION_FAILURE
      if (valid && isV) // h = S D^{-1/2} (c - delta o v)
        sfor<0, N>([&](auto ii) { mine[N * N + N + decltype(ii)::value] = X[decltype(ii)::value]; });
      if (valid && isM)
        sfor<0, N>([&](auto ii) { mine[c * N + decltype(ii)::value] = X[decltype(ii)::value]; });
      asm volatile("" ::: "memory");
    }
  };
  // (Round 3: the blocks of a step loaded COALESCED -- lane c: elements c, c + 16, ... of the contiguous block -- parked in
  // a per-problem LDS region and the lanes' columns read from there, instead of 40-60 per-lane column loads that touch a
  // different cache line in almost every lane: 0.65 / 0.66 / 0.65 ms against 0.58 / 0.66 / 0.62.  The column loads hit
  // in the vector L1 and were not what the top of a step costs; the LDS round trip is added latency.)
  // (Round 3, after the loads became selects: a non-live child's S column, t and v requested at the top of the step
  // with the other inputs instead of inside the step -- 0.82 / 0.93 / 0.85 ms against 0.57 / 0.66 / 0.61: the 27 extra
  // live values spill.)
  // (Requesting the blocks of step s + 1 into a second REGISTER set before the arithmetic of step s did
  // not pay -- 0.87 against 0.84 ms on the heterogeneous chain of the reference's benchmark family: the
  // kernel is at 256 VGPRs + AGPR copies already -- and neither did requesting a non-live child's spill
  // slot with the step's other inputs (+40 registers: 0.84 against 0.78).  The register-free L2 prefetch
  // below does: 0.835 -> 0.775 ms there, 0.96 -> 0.89 / 0.925 -> 0.86 ms on the two trees.)
  // (the record of step s + 1 is read during step s: its scalar loads are not waited for at the top of a step)
#ifdef SIP_TREE_STAMPS
  // segments: 0 step record + issue of the fetch | 1 wait for the fetched blocks | 2 prefetch issue | 3 edge step
  // arithmetic (live child) | 4 edge step (child from the spill) | 5 node step | 6 rollout: loads | 7 rollout: arithmetic
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, seg_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg_last)::"memory");
#endif
  TreeStep nx = ts.backward[0];
  for (int s = 0; s < ts.n_backward; ++s) {
    const TreeStep st = kPrefetch ? nx : ts.backward[s];
    Pre pre;
    fetch(st, pre);
    TREE_SEG(0);
    TREE_SEG_DRAIN();
    TREE_SEG(1);
    if (kPrefetch && s + 1 < ts.n_backward) { // the read-only inputs of the next step, on their way to L2 meanwhile
      nx = ts.backward[s + 1];
      if (nx.kind == 0)
        prefetch(in + nx.oA, (long)nx.nc * (nx.n + nx.m) + (long)nx.n * nx.m + (long)nx.m * nx.m + nx.m, G2{});
      // node block [Q | q | c | delta] of the step's node (an edge step needs it when it opens the node;
      // its child's delta sits in a block an earlier step already read)
      if (nx.kind == 1 || (nx.flags & TS_LOAD_V))
        prefetch(in + nx.oQ, (long)nx.n * (nx.n + 3), G2{});
      // a child finished longer ago comes back from the spill (its S | g | h | t | v slot)
      if (nx.kind == 0 && !(nx.flags & TS_CHILD_LIVE))
        prefetch(pw + (long)nx.child * L::WS, L::WS, GW{});
    }
    TREE_SEG(2);
    backward_step(st, pre);
#ifdef SIP_TREE_STAMPS
    TREE_SEG(st.kind == 1 ? 5 : (st.flags & TS_CHILD_LIVE) ? 3 : 4);
#endif
  }
  // root (the last node step): g = v + W (c - delta o v)  (lqr.cpp:798-819)
  {
    double F[N];
    sfor<0, N>([&](auto ii) { F[decltype(ii)::value] = isV ? Vc[decltype(ii)::value] : 0.0; });
    rank1x<N, N, true>(F, W, tv);
    if (valid && isV) {
      double *gn = pw + (long)ts.root * L::WS + N * N;
      sfor<0, N>([&](auto ii) { gn[decltype(ii)::value] = F[decltype(ii)::value]; });
    }
  }
  if (valid && c == 0)
    status[p] = stat;
  // the rollout reads S / g / h / K / k written above by other lanes of this wave
  __syncthreads();

  // ---- forward rollout (lqr.cpp:821-870); lane r < N owns row r -----------------------------------
  auto sum4 = [](const double (&a)[4]) { return (a[0] + a[1]) + (a[2] + a[3]); };
  double xlast = 0.0; // x of the child the previous step produced
  {                   // x_root = D^{1/2} h_root, y_root = g_root
    const double *slot = pw + (long)ts.root * L::WS;
    const int n = ts.root_n;
    const double dd = load_elem(in + ts.root_od, n, c, true, 1.0);
    const double x0 = (dd * rsqrt_nr(dd)) * slot[N * N + N + cmN];
    if (valid && c < n) {
      out[ts.root_ox + c] = x0;
      out[ts.root_oy + c] = slot[N * N + cmN];
    }
    xlast = c < n ? x0 : 0.0;
  }
  // The operands of a rollout step depend on the step record only, not on x: they are requested ONE STEP AHEAD
  // into a second register set (two sets alternate, the loop runs two steps per trip), while the blocks of the
  // step after that are on their way to L2 (prefetch) and the record of the one after that is being read.  With
  // the loads at the top of the step they serve, the rollout ran at one exposed round trip per step: 37 % of the
  // kernel's cycles for 2 % of arithmetic (tools/tree_stamps.py).
  struct Fwd {
    double KT[N], Arow[N], Brow[M], Wc[N], kk0, gg, hh, dd, xp;
  };
  auto load_fwd = [&](const TreeStep &st, Fwd &o) {
    const int n = st.n, nc = st.nc, m = st.m;
    const double *gi = pg + (long)st.edge * L::GAIN;
    const double *slot = pw + (long)st.child * L::WS;
    const bool anyA = nc > 0 && n > 0, anyB = nc > 0 && m > 0;
    const double *Ar = (anyA ? in + st.oA : in) + (anyA && c < nc ? c : 0); // row c of A (nc x n)
    const double *Br = (anyB ? in + st.oB : in) + (anyB && c < nc ? c : 0); // row c of B (nc x m)
    sfor<0, N>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      o.KT[k] = gi[k * M + cuM];   // padded gains: zeros on the padding
      o.Wc[k] = slot[cmN * N + k]; // S symmetric: row c = column c
      const double al = Ar[(long)(anyA ? nc : 0) * ((anyA & (k < n)) ? k : 0)];
      const bool inA = (k < n) & (c < nc);
      o.Arow[k] = inA ? al : 0.0;
    });
    sfor<0, M>([&](auto jj) {
      constexpr int q = decltype(jj)::value;
      const double bl = Br[(long)(anyB ? nc : 0) * ((anyB & (q < m)) ? q : 0)];
      const bool inB = (q < m) & (c < nc);
      o.Brow[q] = inB ? bl : 0.0;
    });
    o.kk0 = gi[N * M + cuM], o.gg = slot[N * N + cmN], o.hh = slot[N * N + N + cmN];
    o.dd = load_elem(in + st.odc, nc, c, true, 1.0);
    // x of the parent as this lane wrote it when the parent was rolled out (at least two steps ago unless the step
    // is TS_CHILD_LIVE, which takes x from the previous step's registers and ignores this)
    const double xl = out[(c < n) ? st.oxp + c : 0]; // (out_len > 0 is guaranteed by the host)
    o.xp = (c < n) ? xl : 0.0;
  };
  auto prefetch_fwd = [&](const TreeStep &st) { // A | B, gains, spill slot and delta of an edge / its child
    prefetch(in + st.oA, (long)st.nc * (st.n + st.m), G2{});
    prefetch(pg + (long)st.edge * L::GAIN, L::GAIN, G1{});
    prefetch(pw + (long)st.child * L::WS, L::WS, GW{});
    prefetch(in + st.odc, st.nc, G1{});
  };
  auto forward_step = [&](const TreeStep &st, const Fwd &f) {
    const int nc = st.nc, m = st.m;
    const double x = (st.flags & TS_CHILD_LIVE) ? xlast : f.xp;
    const double sdi = rsqrt_nr(f.dd), sdv = f.dd * sdi;
    double acc[4] = {f.kk0, 0.0, 0.0, 0.0};
    dotv<N, true>(acc, x, f.KT);
    const double u = sum4(acc); // u = k + K x  (lqr.cpp:856-857)
    double az[4] = {0.0, 0.0, 0.0, 0.0};
    dotv<N, true>(az, x, f.Arow);
    dotv<M, true>(az, u, f.Brow);
    const double zeta = sum4(az) * sdi; // D^{-1/2} (A x + B u)
    double as[4] = {0.0, 0.0, 0.0, 0.0};
    dotv<N, true>(as, zeta, f.Wc);
    const double sz = sum4(as);
    const double xc = sdv * (sz + f.hh);                   // x_c = D^{1/2} (S zeta + h)
    const double yc = __builtin_fma(sdi, zeta - sz, f.gg); // y_c = g_c + D^{-1/2} (zeta - S zeta)
    if (valid) {
      if (c < m)
        out[st.ou + c] = u;
      if (c < nc) {
        out[st.oxc + c] = xc;
        out[st.oyc + c] = yc;
      }
    }
    xlast = c < nc ? xc : 0.0;
  };
  const int nf = ts.n_forward;
  auto record = [&](const int s) { return ts.forward[s < nf ? s : nf - 1]; };
  if (nf > 0) {
    TreeStep r0 = record(0), r1 = record(1), r2 = record(2);
    Fwd fa, fb;
    load_fwd(r0, fa);
    for (int s = 0; s < nf; s += 2) {
      // step s on set a; r0, r1, r2 = the records of steps s, s + 1, s + 2
      const TreeStep r3 = record(s + 3);
      if (s + 1 < nf)
        load_fwd(r1, fb);
      if (kPrefetch && s + 2 < nf)
        prefetch_fwd(r2);
      TREE_SEG(6);
      forward_step(r0, fa);
      TREE_SEG(7);
      if (s + 1 < nf) { // step s + 1 on set b
        const TreeStep r4 = record(s + 4);
        if (s + 2 < nf)
          load_fwd(r2, fa);
        if (kPrefetch && s + 3 < nf)
          prefetch_fwd(r3);
        TREE_SEG(6);
        forward_step(r1, fb);
        TREE_SEG(7);
        r0 = r2, r1 = r3, r2 = r4;
      }
    }
  }
#ifdef SIP_TREE_STAMPS
  if (lane == 0)
    for (int k = 0; k < 8; ++k)
      atomicAdd(&g_tree_seg[k], seg[k]);
#endif
}

} // namespace sipamd
