// chain_qw16.hpp -- fp64 batched chain Riccati kernel, "quarter-wave" layout.
//
// One problem per 16-lane DPP row (4 problems per wavefront).  Lane c < N of a
// row owns COLUMN c of every n x n / m x n matrix of its problem (rows live in
// registers); lane N carries the affine ("vector") column, so the backward
// affine sweep of LQR::solve (lqr.cpp:738-796: g, h, k, v) rides along in the
// very same instructions that run the matrix recursion of
// LQR::factor_with_status (lqr.cpp:645-731): [F|g] = W [A|t], [H|h] = [M^T|r] +
// B^T [F|g], [K|k] = -G^{-1} [H|h], [V|v] = [Q|q] + A^T [F|g] + K^T [H|h].
//
// Cross-lane traffic is exclusively `v_fmac_f64_dpp ... row_newbcast:k`
// (gfx90a+ DP-ALU DPP): a fused "broadcast lane k of my row, multiply,
// accumulate" at the plain fp64 FMA issue rate -- no LDS round trip, no
// ds_bpermute.  A p x q x r product costs p*q (or q*r) wave instructions for
// four problems at once.
//
// Cholesky (Eigen::LLT of lqr.cpp:505,697) is right-looking on the full
// symmetric storage: by symmetry lane j already holds row k of the trailing
// matrix in its own register k, so the rank-1 update needs only the broadcast
// of the scaled pivot column.  Pivot test d <= 0 reproduces Eigen's
// NumericalIssue condition; statuses follow lqr.hpp:68-74 in the reference's
// order (G before delta before F at a node; first failing node in postorder).
//
// The forward rollout (lqr.cpp:821-870) is restated on S = F^{-1} (the inverse
// that compute_regularized_W forms, lqr.cpp:516-519) with z = A x + B u and
// zeta = D^{-1/2} z:
//   x_c = (I + Delta V)^{-1} (z + c - delta o v) = D^{1/2} (S zeta + h),
//   y_c = g_c + W_c z                            = g_c + D^{-1/2} (zeta - S zeta),
// where g_c = v_c - W_c(delta_c o v_c - c_c) is the reference's own `g`
// (lqr.cpp:778-781) and h = S D^{-1/2} (c_c - delta_c o v_c) is one more
// right-hand side of the F solve of the node (carried by the vector lane), so
// the factor state spilled per node is [S | g | h].  Both lines are the
// reference's formulas (F_inv_mult_vector, lqr.cpp:531-549; W of
// lqr.cpp:521-528) and keep their accuracy for delta anywhere in
// [1e-8, 1e9]: the cheaper x_c = z + c - delta o y_c cancels for large delta.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "dpp_blocks_gen.hpp"

namespace sipamd {

template <int I, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, E>(f);
  }
}
// I, I-1, ..., E+1
template <int I, int E, class F>
__device__ __forceinline__ void sfor_down(F &&f) {
  if constexpr (I > E) {
    f(std::integral_constant<int, I>{});
    sfor_down<I - 1, E>(f);
  }
}

// Value of x in lane K of the caller's 16-lane row (compiler-scheduled
// v_mov_b64_dpp; hipcc pads the DPP hazards of its own instructions).
template <int K>
__device__ __forceinline__ double bcast(double x) {
  double und;
  asm volatile("" : "=v"(und));
  return __builtin_amdgcn_update_dpp(und, x, 0x150 + K, 0xf, 0xf, false);
}

// 1/sqrt(d): v_rsq_f64 seed + one third-order Newton step (full fp64).
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double s = d * y;
  const double e = __builtin_fma(-s, y, 1.0);
  const double t = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(y * e, t, y);
}

// 1/d: v_rcp_f64 seed + one third-order Newton step (full fp64).
__device__ __forceinline__ double rcp_nr(double d) {
  const double y = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y, 1.0);
  const double t = __builtin_fma(e, e, e);
  return __builtin_fma(y, t, y);
}

// In-place square-root-free Cholesky (A = Lt D^-1 Lt^T) of the S x S symmetric
// matrix held one column per lane (lanes c < S).  On exit lane j holds
// Lt(i,j) = L(i,j) * L(j,j) in A[i], i >= j (so A[j] of lane j is the pivot
// d_j = L(j,j)^2 of Eigen's LLT) and dinv[k] = 1 / d_k replicated in every
// lane.  Pivots are those of the Cholesky factorisation, so "a pivot <= 0"
// (Eigen: NumericalIssue) is detected identically.  Returns true on failure.
//
// Step k broadcasts the *unscaled* pivot column (last written by the previous
// step's update, many instructions ago), so no DPP wait state is needed.
template <int S>
__device__ __forceinline__ bool chol_ldl_dpp(double (&A)[S], double (&dinv)[S],
                                             const int c) {
#ifdef SIP_QW16_CHOL_SELECT // the round-1 form: a ballot + scalar OR per pivot, the update column by two selects
  // The pivot is replicated over its row, so the comparison result is
  // row-uniform: collect it as a wave mask (one v_cmp + one s_or per pivot).
  unsigned long long failmask = 0;
  sfor<0, S>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    const double d = bcast<k>(A[k]);
    failmask |= __ballot(d <= 0.0);
    const double y2 = rcp_nr(d);
    dinv[k] = y2;
    // lane j > k: Lt(j,k) / d_k (by symmetry its own A[k] is Lt(j,k))
    const double upd = (c > k && c < S) ? A[k] * y2 : 0.0;
    // A(i,j) -= Lt(i,k) Lt(j,k) / d_k, i > k, lanes j > k
    rank1<S - k - 1, k, true, false>(A + k + 1, A + k + 1, upd);
  });
  return ((failmask >> (__lane_id() & 63)) & 1ull) != 0;
#else
  // The smallest pivot so far (row-replicated like the pivots): one v_min_f64 per pivot and one
  // comparison at the end say "some pivot <= 0".  v_min_f64 returns the other operand for a NaN,
  // so a NaN pivot passes as it passes Eigen's `x <= 0` test.
  double dmin = __builtin_inf();
  sfor<0, S>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    const double d = bcast<k>(A[k]);
    asm("v_min_f64 %0, %1, %2" : "=v"(dmin) : "v"(dmin), "v"(d));
    // lane j > k: Lt(j,k) (by symmetry its own A[k]), zero on the finished columns -- a multiply by
    // a loop-invariant 0/1 that does not wait for the reciprocal
    const double own = A[k] * ((c > k && c < S) ? 1.0 : 0.0);
    const double y2 = rcp_nr(d);
    dinv[k] = y2;
    const double upd = own * y2; // Lt(j,k) / d_k
    // A(i,j) -= Lt(i,k) Lt(j,k) / d_k, i > k, lanes j > k
    rank1<S - k - 1, k, true, false>(A + k + 1, A + k + 1, upd);
  });
  return dmin <= 0.0;
#endif
}

// X <- (Lt D^-1 Lt^T)^{-1} X for X held one column per lane (any lane of the
// row may carry a right-hand side); Lt / dinv from chol_ldl_dpp.
//   forward :  w_j = (x_j - sum_{i<j} Lt(j,i) w_i) / d_j
//   backward:  x_j = w_j - (sum_{i>j} Lt(i,j) x_i) / d_j
template <int S>
__device__ __forceinline__ void ldl_solve_dpp(const double (&Lt)[S],
                                              const double (&dinv)[S],
                                              double (&X)[S]) {
  sfor<0, S>([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    X[j] *= dinv[j];
    rank1<S - j - 1, j, true, (j == 0)>(X + j + 1, Lt + j + 1, X[j]);
  });
  double acc[S];
  sfor<0, S>([&](auto jj) { acc[decltype(jj)::value] = 0.0; });
  sfor_down<S - 1, -1>([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    X[j] = __builtin_fma(-dinv[j], acc[j], X[j]);
    spread<j, false, false>(acc, Lt[j], X[j]); // acc[i] += Lt(j,i) x_j, i < j
  });
}

// SYM: the symmetric-packed layout (SIP_LQR_LAYOUT_SYMMETRIC, include/sip_lqr_amd.h): Q and R are stored as their
// lower triangles packed by columns (column c: rows c .. n-1), n (n + 1) / 2 and m (m + 1) / 2 scalars -- the
// reference reads both triangles of a symmetric Q (lqr.cpp:658) and only the lower one of R (Eigen::LLT, :697), so
// the strictly upper parts are redundant bytes of a bandwidth-bound kernel (576 of 7 840 B per problem-stage at C3).
template <int N, int M, bool SYM = false>
struct ChainLayout {
  // scalars per stage block (see include/sip_lqr_amd.h, "Packed chain layout")
  static constexpr int QLEN = SYM ? N * (N + 1) / 2 : N * N;
  static constexpr int RLEN = SYM ? M * (M + 1) / 2 : M * M;
  static constexpr int NODE = QLEN + N;                  // Q | delta
  static constexpr int EDGE = N * N + 2 * N * M + RLEN;  // A | B | M | R
  static constexpr int VNODE = 2 * N;                  // q | c   (x | y)
  static constexpr int VEDGE = M;                      // r       (u)
  static constexpr int GAIN = M * N + M;               // K | k
  static constexpr int WSN = N * N + 2 * N;            // S | g | h (workspace)
};

// F_factor / W of one node (lqr.cpp:487-529).  V: column c of V (lanes < N).
// dl: delta_c one per lane (1.0 on lanes >= N).  tv: c - delta o v on the
// vector lane, zeros elsewhere.  X returns S = F^{-1} (lanes < N) and, on the
// vector lane, S D^{-1/2} tv.  Returns pivot failure.
// EXPORT (the fused tree kernel's LQR::Workspace outputs): Lt returns the factor of F as chol_ldl_dpp leaves it.
template <int N, bool EXPORT = false>
__device__ __forceinline__ bool node_factor(const double (&V)[N], const double dl,
                                            const int c, const double (&E)[N],
                                            const double (&tv)[N], double (&W)[N],
                                            double (&X)[N], double *Lt = nullptr) {
  const double sdi = rsqrt_nr(dl); // sqrt_delta_inv, lqr.cpp:482
  const double sd = dl * sdi;      // sqrt_delta,     lqr.cpp:481
  double S[N], A[N], rinv[N];
  sfor<0, N>([&](auto ii) { S[decltype(ii)::value] = 0.0; });
  spread<N, false, true>(S, sd, sd); // S[r] = sd_r sd_c
  sfor<0, N>([&](auto ii) {
    constexpr int r = decltype(ii)::value;
    A[r] = __builtin_fma(S[r], V[r], E[r]); // I + D^1/2 V D^1/2, lqr.cpp:497-503
  });
  const bool fail = chol_ldl_dpp<N>(A, rinv, c); // LLT of lqr.cpp:505
  if constexpr (EXPORT)
    sfor<0, N>([&](auto ii) { Lt[decltype(ii)::value] = A[decltype(ii)::value]; });
  sfor<0, N>([&](auto ii) { X[decltype(ii)::value] = E[decltype(ii)::value]; });
  spreadv<N, false>(X, sdi, tv); // vector lane: D^{-1/2} (c - delta o v), lqr.cpp:539-541
  ldl_solve_dpp<N>(A, rinv, X); // F^{-1} [I | .], lqr.cpp:516-519, 542-545
  sfor<0, N>([&](auto ii) { S[decltype(ii)::value] = 0.0; });
  spread<N, false, true>(S, sdi, sdi);
  sfor<0, N>([&](auto ii) {
    constexpr int r = decltype(ii)::value;
    W[r] = (E[r] - X[r]) * S[r]; // lqr.cpp:521-528
  });
  return fail;
}

// Cache policy of the LDS-DMA streams of the problem inputs (each byte is read
// once per sweep direction): 2 = nt, 0 = default.  nt keeps the once-read
// inputs from evicting what the rollout re-reads (gains, [W | g] spill):
// measured +3.5 % at C3 (A/B in one process, tools/ab.sh).
#ifndef SIP_LQR_NT_SOL
#define SIP_LQR_NT_SOL 0
#endif
#ifndef SIP_LQR_NT_IN
#define SIP_LQR_NT_IN 2
#endif
// rollout: re-read of A|B and delta / read of the gains and the spill (last use)
#ifndef SIP_LQR_NT_FAB
#define SIP_LQR_NT_FAB SIP_LQR_NT_IN
#endif
#ifndef SIP_LQR_NT_FSP
#define SIP_LQR_NT_FSP 0
#endif
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(3))) double lds_cdouble;

// Diagnostic build only (-DSIP_LQR_STAMPS, tools/diag_build.sh): shader-clock
// stamps around the phases of a wave; the product build executes none.
#ifdef SIP_LQR_STAMPS
#define SIP_STAMP(var)                                                         \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define SIP_STAMP_ARG , unsigned long long *__restrict__ stamps
// per-segment accumulation: seg[k] += now - last; last = now
#define SIP_SEG(k)                                                             \
  do {                                                                         \
    unsigned long long now_;                                                   \
    SIP_STAMP(now_);                                                           \
    seg[k] += now_ - seg_last;                                                 \
    seg_last = now_;                                                           \
  } while (0)
#else
#define SIP_SEG(k)                                                             \
  do {                                                                         \
  } while (0)
#define SIP_STAMP(var)                                                         \
  do {                                                                         \
  } while (0)
#define SIP_STAMP_ARG
#endif

// LDS-DMA staging of one contiguous per-problem region (PIECES 16-byte pieces
// per problem) for the 4 problems of a wave: `global_load_lds_dwordx4` writes
// LDS as wave-uniform base + lane*16, so instruction j fills image bytes
// [1024 j, 1024 j + 1024) and each lane picks the global piece that belongs
// there.  Image: problem rr of the wave at byte rr * PIECES * 16; the image is padded to whole
// instructions (INSTR * 1024 bytes).
template <int PIECES>
struct StageDma {
  static constexpr int INSTR = (4 * PIECES + 63) / 64;
  static constexpr int BYTES = INSTR * 1024;
  static constexpr int ROW_BYTES = PIECES * 16; // problem rr's image starts at rr * ROW_BYTES
  // The instruction's immediate offset is added to the global AND to the LDS address, so four
  // consecutive instructions (LDS images 1 KiB apart) share one M0 value: instruction j carries
  // the immediate (j % 4) * 1024, its lanes' global offsets are lowered by as much, and M0 moves
  // once per group of four (one s_add + its wait state instead of four).  BIAS keeps the lowered
  // offsets non-negative (they are zero-extended into the address).
#ifdef SIP_QW16_DMA_PLAIN
  static constexpr int GROUP = 1;
#else
  static constexpr int GROUP = 4;
#endif
  static constexpr unsigned BIAS = (GROUP - 1) * 1024u;
  unsigned off[INSTR > 0 ? INSTR : 1];
  __device__ __forceinline__ void init(const int lane,
                                       const unsigned problem_stride_bytes,
                                       const unsigned max_rel) {
    sfor<0, INSTR>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      // lanes past the last piece re-read it (their LDS slots are padding of the image)
      const unsigned q = (unsigned)(j * 64 + lane) < 4u * PIECES ? (unsigned)(j * 64 + lane) : 4u * PIECES - 1u;
      unsigned rr = q / PIECES;
      const unsigned within = q - rr * PIECES;
      rr = rr < max_rel ? rr : max_rel;
      off[j] = rr * problem_stride_bytes + within * 16u + BIAS - (unsigned)(j % GROUP) * 1024u;
    });
  }
  // base: wave-uniform pointer to the region of the wave's first problem.
  // AUX: cache-policy bits of the load (0 default, 2 = nt for once-read bytes).
  template <int AUX = 0>
  __device__ __forceinline__ void issue(const char *base, lds_char *dst,
                                        const int lane) const {
    // Every instruction is issued by the whole wave, never under a lane predicate: two
    // identically predicated tails in a row (e.g. 4 * 305 and 4 * 17 pieces: lanes < 4 both) were
    // tail-merged by the compiler into ONE instruction whose LDS base -- a wave-uniform M0 value --
    // became a per-lane PHI read with v_readfirstlane, which scattered one image into the other.
    const char *lowered = base - BIAS;
    sfor<0, INSTR>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(lowered + off[j]),
                                       (__attribute__((address_space(3))) void *)(dst + (j - j % GROUP) * 1024), 16,
                                       (j % GROUP) * 1024, AUX);
    });
  }
};

// The same lane-linear staging in 4-byte pieces (`global_load_lds_dword`, 256 bytes per instruction), for
// regions whose length is an odd number of scalars (the gains K | k when M is odd): no byte past the
// region is read -- a 16-byte piece would reach 8 bytes past the last problem's last stage.
template <int DWORDS>
struct StageDmaDwords {
  static constexpr int INSTR = (4 * DWORDS + 63) / 64;
  static constexpr int BYTES = (INSTR * 256 + 1023) / 1024 * 1024;
  static constexpr int ROW_BYTES = DWORDS * 4;
  unsigned off[INSTR > 0 ? INSTR : 1];
  __device__ __forceinline__ void init(const int lane, const unsigned problem_stride_bytes, const unsigned max_rel) {
    sfor<0, INSTR>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      const unsigned q = (unsigned)(j * 64 + lane) < 4u * DWORDS ? (unsigned)(j * 64 + lane) : 4u * DWORDS - 1u;
      unsigned rr = q / DWORDS;
      const unsigned within = q - rr * DWORDS;
      rr = rr < max_rel ? rr : max_rel;
      off[j] = rr * problem_stride_bytes + within * 4u;
    });
  }
  template <int AUX = 0>
  __device__ __forceinline__ void issue(const char *base, lds_char *dst, const int lane) const {
    sfor<0, INSTR>([&](auto jj) { // whole-wave instructions only (see StageDma::issue)
      constexpr int j = decltype(jj)::value;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off[j]),
                                       (__attribute__((address_space(3))) void *)(dst + j * 256), 4, 0, AUX);
    });
  }
};

// The same staging with every instruction inside ONE problem's region: the image of problem rr
// starts at rr * IP KiB (IP = instructions per problem, the region padded to whole instructions)
// and instruction (rr, jj) moves its pieces [64 jj, 64 jj + 64).  All lanes then use the same byte
// offset 16 * lane for every instruction (one more for the last instruction of a problem, whose
// lanes past the last piece re-read it), the problem's base is wave-uniform, and jj rides in the
// immediate offset: 2 offset registers instead of one (64-bit) per instruction, one address add
// per problem and one M0 value per four instructions.
template <int PIECES>
struct StageDmaRows {
  static constexpr int IP = (PIECES + 63) / 64;
  static constexpr int INSTR = 4 * IP;
  static constexpr int ROW_BYTES = IP * 1024;
  static constexpr int BYTES = 4 * ROW_BYTES;
  static constexpr bool RAGGED = PIECES % 64 != 0;
  unsigned vo_main, vo_tail;
  unsigned long long row[4]; // wave-uniform byte offset of problem rr's region (clamped at the batch end)
  __device__ __forceinline__ void init(const int lane,
                                       const unsigned problem_stride_bytes,
                                       const unsigned max_rel) {
    vo_main = (unsigned)lane * 16u;
    const unsigned last = (unsigned)(PIECES - 1 - 64 * (IP - 1)); // last piece of the last instruction
    vo_tail = ((unsigned)lane < last ? (unsigned)lane : last) * 16u;
    sfor<0, 4>([&](auto rq) {
      constexpr unsigned r = decltype(rq)::value;
      row[r] = (unsigned long long)(r < max_rel ? r : max_rel) * problem_stride_bytes;
    });
  }
  template <int AUX = 0>
  __device__ __forceinline__ void issue(const char *base, lds_char *dst,
                                        const int lane) const {
    // whole-wave instructions only (see StageDma::issue)
    sfor<0, 4>([&](auto rq) {
      constexpr int r = decltype(rq)::value;
      const char *region = base + row[r];
      sfor<0, IP>([&](auto jq) {
        constexpr int jj = decltype(jq)::value;
        constexpr int g = jj / 4; // immediates reach 3 KiB: a new base every four instructions
        const unsigned vo = (RAGGED && jj == IP - 1) ? vo_tail : vo_main;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(region + g * 4096 + vo),
            (__attribute__((address_space(3))) void *)(dst + r * ROW_BYTES + g * 4096), 16, (jj % 4) * 1024, AUX);
      });
    });
  }
};

// Byte sizes of the LDS images of the staged kernel (one wave per block).
// SPLIT (the Newton-KKT step, sip_lqr_factor_solve_split): A | B are not part of the mats stage
// block -- [Q | delta | M | R] is -- and stream from a second array (the dynamics Jacobians where the
// model callback left them), as an image of their own behind the mats image.
template <int N, int M, bool WPACK, bool SPLIT = false, bool SYM = false>
struct StagedCfg {
  using L = ChainLayout<N, M, SYM>;
  static_assert(!SYM || (L::QLEN % 2 == 0 && L::RLEN % 2 == 0 && N % 2 == 0 && M % 2 == 0),
                "the symmetric-packed kernels are instantiated where every block stays a whole number of 16-byte pieces");
  static constexpr int AB = N * N + N * M;                      // A | B
  static constexpr int STG = L::NODE + L::EDGE - (SPLIT ? AB : 0); // mats stage stride
  static constexpr int WSN = // S | g | h, even number of scalars
      WPACK ? ((N * (N + 1) / 2 + 2 * N + 1) / 2) * 2 : L::WSN;
  // backward: whole stage block of mats + of vecs.  The per-problem form of the mats stream pads
  // every problem to whole instructions; it is taken unless that padding costs a wavefront per CU
  // (160 KiB of LDS: four wavefronts need <= 40 KiB each).
  // (an odd number of scalars -- M odd -- takes one more 16-byte piece, whose second half is the first
  // scalar of the block behind it: the next stage's, or the terminal node's; sources are then only
  // 8-byte aligned, which the LDS-DMA of gfx950 copies exactly, tools/ubench/lds_dma_align.hip)
  using BV = StageDma<(L::VNODE + L::VEDGE + 1) / 2>;
  using BMflat = StageDma<(STG + 1) / 2>;
  using BMrows = StageDmaRows<(STG + 1) / 2>;
  struct NoStream { // the A | B image of the unsplit kernel: none
    static constexpr int INSTR = 0, BYTES = 0, ROW_BYTES = 0;
  };
  static_assert(!SPLIT || AB % 2 == 0, "the split kernels are instantiated for even N (A | B in whole 16-byte pieces)");
  using BA = std::conditional_t<SPLIT, StageDmaRows<AB / 2>, NoStream>;
  static constexpr int SCR_BYTES_ = ((4 * 2 * N + N) * 8 + 1023) / 1024 * 1024;
  static constexpr int waves_per_cu(int bm_bytes) {
    const int per_wave = 2 * (bm_bytes + BA::BYTES + BV::BYTES) + SCR_BYTES_;
    const int w = 163840 / per_wave;
    return w > 4 ? 4 : w;
  }
#ifdef SIP_QW16_DMA_FLAT
  using BM = BMflat;
#else
  using BM = std::conditional_t<SPLIT || waves_per_cu(BMrows::BYTES) >= waves_per_cu(BMflat::BYTES), BMrows, BMflat>;
#endif
  static constexpr int B_BYTES = BM::BYTES + BA::BYTES + BV::BYTES;
  // forward: A|B, gains, S|g|h of the child, delta of the child
  using FA = StageDma<(N * N + N * M + 1) / 2>; // (odd: N odd, M even -- its last half piece is M's first scalar)
  using FG = std::conditional_t<L::GAIN % 2 == 0, StageDma<L::GAIN / 2>, StageDmaDwords<2 * L::GAIN>>;
  using FW = StageDma<WSN / 2>;
  // delta: the last N scalars of the terminal node end the problem's mats -- odd N streams them as dwords
  using FC = std::conditional_t<N % 2 == 0, StageDma<N / 2>, StageDmaDwords<2 * N>>;
  static constexpr int F_BYTES = FA::BYTES + FG::BYTES + FW::BYTES + FC::BYTES;
  // LDS-DMA instructions per forward stage: the count the rollout's
  // `s_waitcnt vmcnt(F_GLDS)` relies on.
  static constexpr int F_GLDS = FA::INSTR + FG::INSTR + FW::INSTR + FC::INSTR;
  static constexpr int B_NBUF = 2; // backward: one stage ahead
  // forward: F_NBUF - 1 stages ahead.  Three buffers (two stages of lookahead) unless the third one
  // costs a wavefront per CU (160 KiB of LDS): there the occupancy is worth more than the lookahead --
  // (12,6) 0.47 -> 0.36 ms, (13,5) 0.51 -> 0.40, (12,8) 0.52 -> 0.42, (14,4) 0.54 -> 0.44 at batch 4096.
  static constexpr int lds_main(int fbufs) {
    return B_NBUF * B_BYTES > fbufs * F_BYTES ? B_NBUF * B_BYTES : fbufs * F_BYTES;
  }
  static constexpr int waves_with(int fbufs) {
    const int w = 163840 / (lds_main(fbufs) + SCR_BYTES_);
    return w > 4 ? 4 : w;
  }
#ifdef SIP_LQR_FNBUF
  static constexpr int F_NBUF = SIP_LQR_FNBUF;
#else
  static constexpr int F_NBUF = (waves_with(2) > waves_with(3) || lds_main(3) + SCR_BYTES_ > 65536) ? 2 : 3; // (or it would not fit at all)
#endif
  static constexpr int LDS_MAIN = lds_main(F_NBUF);
  // per-row scratch behind the stage buffers: t (N) | v_child (N) of the
  // vector lane, then one block of N zeros
  static constexpr int SCR_BYTES = ((4 * 2 * N + N) * 8 + 1023) / 1024 * 1024;
  static constexpr int LDS_BYTES = LDS_MAIN + SCR_BYTES;
  static constexpr bool OK = N <= 16;
  static constexpr bool WIDE = N % 2 == 0; // columns of n scalars are 16-byte aligned in the images: 16-byte LDS accesses
};

// Fused factor + solve.  STAGED: stage blocks travel HBM -> LDS by LDS-DMA one
// stage ahead of the arithmetic (needs N and M even: 16-byte pieces);
// otherwise every lane loads its columns straight from global memory.
// WPACK: spill only the lower triangle of the symmetric W.
// SPLIT (staged, mode 0 only): stage i's A | B of problem p are the N * (N + M) scalars at
// ab + p * ab_pstride + i * ab_sstride (column-major A then B, as in the packed layout) and the
// mats stage block is [Q | delta | M | R].
template <int N, int M, bool STAGED, bool WPACK, bool SPLIT = false, bool SYM = false>
__global__ __launch_bounds__(64) void chain_factor_solve_qw16(
    const double *__restrict__ mats, const double *__restrict__ vecs,
    double *__restrict__ sol, double *__restrict__ gains,
    double *__restrict__ wsp, int *__restrict__ status, const long batch,
    const int T, const int mode, double *__restrict__ gfac,
    const double *__restrict__ ab, const long ab_pstride, const long ab_sstride SIP_STAMP_ARG) {
  // mode 0: fused factor + solve.  mode 1 (split sip_lqr_factor): the backward sweep only, and the
  // LDL factors of the G matrices go to `gfac` ([problem][stage][Lt (M x M, column per lane) | 1/d
  // (M)]).  mode 2 (split sip_lqr_solve): no matrix work at all -- the affine sweep of
  // LQR::solve (lqr.cpp:738-796) runs on distributed vectors against the factor state a mode-1
  // launch left (S in the spill, K in the gains, the G factors in gfac), then the rollout.
  // N = 16 fills the 16-lane row: no lane is left for the affine column, so the vectors of
  // LQR::solve's backward sweep are kept DISTRIBUTED (lane r holds element r) and every
  // matrix-vector product of the sweep is one dotv block (VDIST mode; direct loads only).
  constexpr bool VDIST = N == 16;
  static_assert(N >= 1 && N <= 16, "one problem per 16-lane row");
  static_assert(M >= 1 && M <= 16, "");
  using L = ChainLayout<N, M, SYM>;
  using C = StagedCfg<N, M, WPACK, SPLIT, SYM>;
  static_assert(!SPLIT || (STAGED && WPACK), "the split kernel is the staged one");
  static_assert(!SYM || (STAGED && !VDIST), "the symmetric-packed layout is served by the staged kernels");
  unsigned long long ts_begin = 0, ts_term = 0, ts_bwd = 0, ts_root = 0,
                     ts_end = 0, ts_a = 0, ts_b = 0, acc_bwait = 0,
                     acc_fwait = 0;
  (void)ts_begin, (void)ts_term, (void)ts_bwd, (void)ts_root, (void)ts_end,
      (void)ts_a, (void)ts_b, (void)acc_bwait, (void)acc_fwait;
  unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, seg_last = 0;
  (void)seg, (void)seg_last;
  SIP_STAMP(ts_begin);
  static_assert(!STAGED || C::LDS_BYTES <= 65536, "the LDS images of this shape exceed a workgroup's 64 KiB");
  constexpr int STG = C::STG;               // mats stage stride
  // M^T and R inside a mats stage block (A | B sit between the node part and them unless SPLIT)
  constexpr int OFF_M = L::NODE + (SPLIT ? 0 : N * N + N * M), OFF_R = OFF_M + N * M;
  constexpr int VSTG = L::VNODE + L::VEDGE; // vecs / sol stage stride
  constexpr int WSN = C::WSN;
  constexpr int WG = WPACK ? N * (N + 1) / 2 : N * N; // offset of g in a slot

  extern __shared__ double lds_raw[];
  lds_char *const lds = (lds_char *)lds_raw;

  const int lane = threadIdx.x & 63;
  const int c = lane & 15;
  const int rr = lane >> 4;
  const long p0 = (long)blockIdx.x * 4;
  long p = p0 + rr;
  const bool valid = p < batch;
  if (!valid)
    p = batch - 1;
  const bool isM = c < N;
  const bool isV = c == N;
  const int cm = isM ? c : N - 1;   // clamped matrix column / row
  const int cu = c < M ? c : M - 1; // clamped control column / row

  const long mats_len = (long)(T + 1) * L::NODE + (long)T * (STG - L::NODE);
  const long vecs_len = (long)(T + 1) * L::VNODE + (long)T * L::VEDGE;
  const long gains_len = (long)T * L::GAIN;
  const long ws_len = (long)(T + 1) * WSN;
  const double *pm = mats + p * mats_len;
  const double *pv = vecs + p * vecs_len;
  double *ps = sol + p * vecs_len;
  double *pg = gains + p * gains_len;
  double *pw = wsp + p * ws_len;
  const unsigned max_rel =
      (unsigned)((batch - 1 - p0) < 3 ? (batch - 1 - p0) : 3);

  double E[N];
  sfor<0, N>([&](auto ii) {
    constexpr int r = decltype(ii)::value;
    E[r] = (c == r) ? 1.0 : 0.0;
  });

  // Offsets of S(r, k), k = 0..N-1, r = this lane's row, inside a spill slot (built where the
  // spill is read -- the mode-2 sweep and the rollout -- not carried through the backward sweep).
  auto spill_row_offsets = [&](int(&woff)[N]) {
    sfor<0, N>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      if constexpr (WPACK) {
        const int lo = k < cm ? k : cm, hi = k < cm ? cm : k; // S(hi, lo)
        woff[k] = lo * N - (lo * (lo - 1)) / 2 + (hi - lo);
      } else {
        woff[k] = cm * N + k; // S symmetric: row r = column r
      }
    });
  };

  int stat = 0;
  double W[N], V[N], t[N], vch[N];
  double vd = 0.0, td = 0.0, qd = 0.0; // VDIST: v, t = c - delta o v, q of the current node (element c)
  auto sum4 = [](const double (&a)[4]) { return (a[0] + a[1]) + (a[2] + a[3]); };

  // Buffer view of this wave's rows of the workspace (packed-W spill).
  const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(wsp + p0 * ws_len), 0, (int)((max_rel + 1) * ws_len * 8), 0x00020000);
  // byte offset of W(0, c) of lane c's column inside the wave's view
  // (out of range for lanes that never store)
  const int wstore_base = (valid && isM)
                              ? (int)(rr * ws_len * 8) + (c * N - (c * (c - 1)) / 2 - c) * 8
                              : 0x7ffff000;

  // STAGED: the vector lane parks t = c - delta o v and v (needed by the
  // parent step only on that lane) in LDS; every other lane reads zeros there.
  typedef __attribute__((address_space(3))) double lds_double;
  lds_char *const scr = lds + C::LDS_MAIN;
  lds_double *const my_t = (lds_double *)(scr + rr * (2 * N * 8));
  lds_double *const my_v = my_t + N;
  lds_cdouble *const zeros = (lds_cdouble *)(scr + 4 * (2 * N * 8));
  if constexpr (STAGED) {
    if (lane < N)
      ((lds_double *)zeros)[lane] = 0.0;
  }

  // Loads [Q_i | q_i] as the augmented column.  nm / nv: stage block of mats /
  // vecs (global or LDS).
  auto load_vq = [&](auto nm, auto nv, double(&Vq)[N]) {
    if constexpr (SYM) {
      // column c of the symmetric Q out of its packed lower triangle: Q(r, c) = Q(max, min), packed column `min`
      // starts at min N - min (min - 1) / 2 and holds rows min .. N - 1.  The vector lane reads q (offsets into nv).
      const int base_c = cm * N - (cm * (cm - 1)) / 2 - cm; // + r for r >= c
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        constexpr int base_r = r * N - (r * (r - 1)) / 2 - r; // + c for c > r
        const int off = r >= cm ? base_c + r : base_r + cm;
        auto src = isV ? nv + r : nm + off; // one read per element: the address is selected, not the value
        Vq[r] = *src;
      });
    } else {
    auto src = isV ? nv : nm + cm * N;
    sfor<0, N>([&](auto ii) {
      constexpr int r = decltype(ii)::value;
      Vq[r] = src[r];
    });
    }
    if constexpr (VDIST)
      qd = nv[cm];
  };

  // Common tail of every node: statuses, F/W (lqr.cpp:722-727 + 689), the
  // vector-lane terms for the parent step (t = c - delta o v = -f of
  // lqr.cpp:778-779, and v), and the W spill.
  // Operands of finish_node, fetched ahead of it: delta_c one per lane, and
  // (meaningful on the vector lane) c and delta as columns.
  struct NodeTail {
    double dl, cv[N], dv[N];
    double cd; // VDIST: c_c of the node, one per lane
  };
  auto load_tail = [&](auto nm, auto nv, NodeTail &nt) {
    const double d = nm[L::QLEN + cm];
    nt.dl = isM ? d : 1.0;
    nt.cd = VDIST ? nv[N + cm] : 0.0;
    if constexpr (std::is_same_v<decltype(nm), lds_cdouble *>) {
      auto csrc = isV ? nv + N : zeros, dsrc = isV ? nm + L::QLEN : zeros;
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        nt.cv[r] = csrc[r];
        nt.dv[r] = dsrc[r];
      });
    } else {
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        nt.cv[r] = isV ? nv[N + r] : 0.0;
        nt.dv[r] = isV ? nm[L::QLEN + r] : 0.0;
      });
    }
  };
  auto finish_node = [&](const int i, const NodeTail &nt) {
    const double dl = nt.dl;
    const unsigned long long bad = __ballot(isM && dl <= 0.0);
    const bool bad_row = ((bad >> (lane & 48)) & 0xffffull) != 0;
    if (stat == 0 && bad_row)
      stat = 1; // INVALID_DELTA
    // c - delta o v on the vector lane, exact zeros elsewhere (cv = dv = 0)
    double tv[N], X[N];
    sfor<0, N>([&](auto ii) {
      constexpr int r = decltype(ii)::value;
      tv[r] = nt.cv[r] - nt.dv[r] * V[r];
    });
    if constexpr (STAGED && C::WIDE) {
      if (isV) { // 16-byte pairs (N even): ds_write_b128 at immediate offsets from one address
        typedef double d2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) d2 lds_d2;
        lds_d2 *const t2 = (lds_d2 *)my_t, *const v2 = (lds_d2 *)my_v;
        sfor<0, N / 2>([&](auto kk) {
          constexpr int k = decltype(kk)::value;
          t2[k] = d2{tv[2 * k], tv[2 * k + 1]};
          v2[k] = d2{V[2 * k], V[2 * k + 1]};
        });
      }
    } else if constexpr (STAGED) {
      if (isV) {
        sfor<0, N>([&](auto ii) {
          constexpr int r = decltype(ii)::value;
          my_t[r] = tv[r];
          my_v[r] = V[r];
        });
      }
    } else {
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        t[r] = tv[r];
        vch[r] = V[r];
      });
    }
    if constexpr (VDIST)
      td = nt.cd - dl * vd; // t = c - delta o v (-f of lqr.cpp:778-779), element c
    SIP_SEG(7);
    const bool ffail = node_factor<N>(V, dl, c, E, tv, W, X);
    SIP_SEG(8);
    if (stat == 0 && ffail)
      stat = 2; // F_FACTORIZATION_FAILURE
    if constexpr (VDIST) { // h = S D^{-1/2} (c - delta o v): row c of the symmetric S is this lane's column
      const double phi = rsqrt_nr(dl) * td;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, phi, X);
      if (valid)
        pw[(long)i * WSN + WG + N + c] = sum4(acc);
    }
    if (valid && isV) { // h = S D^{-1/2} (c - delta o v)
      double *hn = pw + (long)i * WSN + WG + N;
      sfor<0, N>([&](auto ii) { hn[decltype(ii)::value] = X[decltype(ii)::value]; });
    }
    if constexpr (WPACK) {
      // Column c of the lower triangle (rows c..N-1), packed by columns.
      // Ragged and lane-dependent, so instead of 12 exec-masked stores the
      // unwanted (row < c, vector lane, tail rows) lanes get an out-of-range
      // buffer offset: the range check of the buffer store drops them.
      sfor<0, N>([&](auto ii) {
        constexpr int r = decltype(ii)::value;
        const int vo = (r >= c) ? wstore_base : 0x7ffff000;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, X[r]), ws_rsrc, vo + r * 8,
                                              i * (WSN * 8), 0);
      });
    } else if (valid && isM) {
      double *wn = pw + (long)i * WSN + c * N;
      sfor<0, N>(
          [&](auto ii) { wn[decltype(ii)::value] = X[decltype(ii)::value]; });
    }
  };

  // One backward step over edge i (lqr.cpp:660-720 and :746-795).  nm / nv:
  // stage block i of mats / vecs (global memory, or its LDS image).  Operands
  // are fetched one segment ahead of their use (the memory fences keep the
  // compiler from hoisting all ~140 registers of a stage to the top).
  auto backward_edge = [&](const int i, auto nm, auto ea, auto nv, NodeTail &nt) { // ea: the stage's A | B

    // [F | g - v_c] = W [A | t]   (lqr.cpp:703 and :780-781)
    double F[N], Aaug[N], Bcol[N];
    double Hc[M], G[M], rinvG[M], H[M], K[M];
    {
      auto msrc = isV ? nv + L::VNODE : nm + (OFF_M + cm);
      sfor<0, M>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if constexpr (SYM) { // R(j, c) = R(max, min) of the packed lower triangle
          const int lo = j < cu ? j : cu, hi = j < cu ? cu : j;
          G[j] = nm[OFF_R + lo * M - (lo * (lo - 1)) / 2 + (hi - lo)];
        } else
        G[j] = nm[OFF_R + cu * M + j]; // column c of R
        // column c of M^T = row c of M; vector lane: r
        H[j] = isV ? msrc[j] : msrc[j * N];
      });
    }
    if constexpr (STAGED && C::WIDE) {
      // 16-byte aligned columns (N even): let the compiler use ds_read_b128
      typedef double d2 __attribute__((ext_vector_type(2)));
      typedef const __attribute__((address_space(3))) d2 lds_cd2;
      lds_cd2 *a_src = (lds_cd2 *)(isV ? (lds_cdouble *)my_t : ea + cm * N);
      lds_cd2 *f_src = (lds_cd2 *)(isV ? (lds_cdouble *)my_v : zeros);
      lds_cd2 *b_src = (lds_cd2 *)(ea + N * N + cu * N);
      sfor<0, N / 2>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const d2 a2 = a_src[k], f2 = f_src[k], b2 = b_src[k];
        Aaug[2 * k] = a2[0], Aaug[2 * k + 1] = a2[1]; // vector lane: t
        F[2 * k] = f2[0], F[2 * k + 1] = f2[1];       // vector lane accumulates g = v_c + W t
        Bcol[2 * k] = b2[0], Bcol[2 * k + 1] = b2[1];
      });
    } else if constexpr (STAGED) { // odd N: 8-byte LDS reads
      lds_cdouble *a_src = isV ? (lds_cdouble *)my_t : ea + cm * N;
      lds_cdouble *f_src = isV ? (lds_cdouble *)my_v : zeros;
      lds_cdouble *b_src = ea + N * N + cu * N;
      sfor<0, N>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        Aaug[k] = a_src[k], F[k] = f_src[k], Bcol[k] = b_src[k];
      });
    } else {
      sfor<0, N>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        Aaug[k] = isV ? t[k] : ea[cm * N + k];
        F[k] = isV ? vch[k] : 0.0;
        Bcol[k] = ea[N * N + cu * N + k];
      });
    }
    rank1x<N, N, true>(F, W, Aaug);
    if (valid && isV) {
      double *gn = pw + (long)(i + 1) * WSN + WG;
      sfor<0, N>(
          [&](auto ii) { gn[decltype(ii)::value] = F[decltype(ii)::value]; });
    }
    double gd = 0.0, hd = 0.0, hf[M], kf[M]; // VDIST: g (element c), h (element c < M), h and k replicated
    if constexpr (VDIST) { // g = v_c + W t  (lqr.cpp:780-781 with t = -f)
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, td, W);
      gd = vd + sum4(acc);
      if (valid)
        pw[(long)(i + 1) * WSN + WG + c] = gd;
    }
    SIP_SEG(2);
    asm volatile("" ::: "memory");
    double Vn[N];
    load_vq(nm, nv, Vn); // [Q | q]: in flight behind the G / K work

    // H_child = B^T W (lqr.cpp:692); G = R + H_child B (lqr.cpp:693-694)
    sfor<0, M>([&](auto jj) { Hc[decltype(jj)::value] = 0.0; });
    spreadx<M, N, false>(Hc, Bcol, W);
    rank1x<M, N, true>(G, Hc, Bcol);
    SIP_SEG(3);
    const bool gfail = chol_ldl_dpp<M>(G, rinvG, c); // lqr.cpp:696-701
    if (mode == 1 && valid) { // G_factor for later solve-only launches (LQR::Workspace::G_factor)
      double *gf = gfac + (p * (long)T + i) * (M * M + M);
      if (c < M)
        sfor<0, M>([&](auto jj) { gf[c * M + decltype(jj)::value] = G[decltype(jj)::value]; });
      if (c == 0)
        sfor<0, M>([&](auto jj) { gf[M * M + decltype(jj)::value] = rinvG[decltype(jj)::value]; });
    }
    SIP_SEG(4);
    if (stat == 0 && gfail)
      stat = 3; // G_FACTORIZATION_FAILURE

    // [H | h] = [M^T | r] + B^T [F | g]   (lqr.cpp:704-705, :783-784)
    spreadx<M, N, false>(H, Bcol, F);
    if constexpr (VDIST) { // h = r + B^T g  (lqr.cpp:783-784); lane j < M holds h_j
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, gd, Bcol);
      hd = nv[L::VNODE + cu] + sum4(acc);
    }
    // [K | k] = -G^{-1} [H | h]   (lqr.cpp:707-713, :785-791)
    sfor<0, M>(
        [&](auto jj) { K[decltype(jj)::value] = H[decltype(jj)::value]; });
    ldl_solve_dpp<M>(G, rinvG, K);
    sfor<0, M>(
        [&](auto jj) { K[decltype(jj)::value] = -K[decltype(jj)::value]; });
    if (valid && c <= N) {
      double *gi = pg + (long)i * L::GAIN + c * M;
      sfor<0, M>(
          [&](auto jj) { gi[decltype(jj)::value] = K[decltype(jj)::value]; });
    }
    if constexpr (VDIST) {
      // k = -G^{-1} h  (lqr.cpp:785-791) on a replicated copy of h: every lane gathers h and the
      // strictly lower part of Lt (lane i holds column i), then runs ldl_solve_dpp's recurrences
      //   w_j = (h_j - sum_{i<j} Lt(j,i) w_i) / d_j,   x_j = w_j - (sum_{i>j} Lt(i,j) x_i) / d_j
      double Lf[M][M];
      sfor<0, M>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        hf[j] = bcast<j>(hd);
        sfor<0, j>([&](auto iv) {
          constexpr int i2 = decltype(iv)::value;
          Lf[j][i2] = bcast<i2>(G[j]); // Lt(j, i), j > i
        });
      });
      sfor<0, M>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        double w = hf[j];
        sfor<0, j>([&](auto iv) { w = __builtin_fma(-Lf[j][decltype(iv)::value], kf[decltype(iv)::value], w); });
        kf[j] = w * rinvG[j];
      });
      sfor_down<M - 1, -1>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        double a = 0.0;
        sfor<j + 1, M>([&](auto iv) { a = __builtin_fma(Lf[decltype(iv)::value][j], kf[decltype(iv)::value], a); });
        kf[j] = __builtin_fma(-rinvG[j], a, kf[j]);
      });
      sfor<0, M>([&](auto jj) { kf[decltype(jj)::value] = -kf[decltype(jj)::value]; });
      if (valid && c == 0) {
        double *gk = pg + (long)i * L::GAIN + N * M;
        sfor<0, M>([&](auto jj) { gk[decltype(jj)::value] = kf[decltype(jj)::value]; });
      }
    }

    SIP_SEG(5);
    // [V | v] = [Q | q] + A^T [F | g] + K^T [H | h]  (lqr.cpp:715-719,:793-794)
    // (Aaug's vector lane is never broadcast: spread reads lanes < N only)
    spreadx<N, N, false>(Vn, Aaug, F);
    asm volatile("" ::: "memory");
    load_tail(nm, nv, nt); // c, delta of the node: in flight behind the K^T H product (F and A are dead by now)
    asm volatile("" ::: "memory");
    spreadx<N, M, true>(Vn, K, H);
    sfor<0, N>(
        [&](auto ii) { V[decltype(ii)::value] = Vn[decltype(ii)::value]; });
    if constexpr (VDIST) { // v = q + A^T g + K^T h  (lqr.cpp:793-794), element c
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, gd, Aaug);
      double kh = 0.0;
      sfor<0, M>([&](auto jj) { kh = __builtin_fma(K[decltype(jj)::value], hf[decltype(jj)::value], kh); });
      vd = qd + sum4(acc) + kh;
    }
    SIP_SEG(6);
    asm volatile("" ::: "memory");
  };

  if (mode != 2) {
  // ---- terminal node (lqr.cpp:651-658 with no child edge) ----------------
  typename C::BM dma_bm;
  typename C::BA dma_ba;
  typename C::BV dma_bv;
  // stage i -> the LDS buffer at `buf`: [mats image | A|B image (SPLIT) | vecs image]
  auto issue_backward = [&](const int i, lds_char *buf) {
    dma_bm.template issue<SIP_LQR_NT_IN>((const char *)(mats + p0 * mats_len + (long)i * STG), buf, lane);
    if constexpr (SPLIT)
      dma_ba.template issue<SIP_LQR_NT_IN>((const char *)(ab + p0 * ab_pstride + (long)i * ab_sstride),
                                           buf + C::BM::BYTES, lane);
    dma_bv.template issue<SIP_LQR_NT_IN>((const char *)(vecs + p0 * vecs_len + (long)i * VSTG),
                                         buf + C::BM::BYTES + C::BA::BYTES, lane);
  };
  if constexpr (STAGED) {
    dma_bm.init(lane, (unsigned)(mats_len * 8), max_rel);
    if constexpr (SPLIT)
      dma_ba.init(lane, (unsigned)(ab_pstride * 8), max_rel);
    dma_bv.init(lane, (unsigned)(vecs_len * 8), max_rel);
    if (T > 0)
      issue_backward(T - 1, lds + ((T - 1) & 1) * C::B_BYTES);
  }
  {
    NodeTail nt;
    load_vq(pm + (long)T * STG, pv + (long)T * VSTG, V);
    if constexpr (VDIST)
      vd = qd; // v_T = q_T
    load_tail(pm + (long)T * STG, pv + (long)T * VSTG, nt);
    finish_node(T, nt);
  }
  SIP_STAMP(ts_term);

  // ---- backward recursion over edges i = T-1 .. 0 -------------------------
  // One stage of the sweep; `par` = i & 1 selects the LDS buffer stage i landed in.  The staged
  // kernel runs the loop two stages per trip with the parity a compile-time constant, so every LDS
  // address of a stage (buffer + row + lane-dependent column offsets) is loop-invariant and leaves
  // the loop instead of being rebuilt by ~50 vector instructions per stage.
  auto stage_step = [&](const int i, auto par) {
    SIP_STAMP(ts_a);
#ifdef SIP_LQR_STAMPS
    seg_last = ts_a;
    if (stamps != nullptr && lane == 0 && i < 64) // per-stage trace behind the per-wave records
      stamps[(long)gridDim.x * 24 + (long)blockIdx.x * 128 + i] = ts_a;
#endif
    if constexpr (STAGED) {
      constexpr int PAR = decltype(par)::value;
      lds_char *buf = lds + PAR * C::B_BYTES;
      // Stage i has landed.  The N buffer stores of the packed-W spill are the
      // last vector-memory operations of the previous node (every lane issues
      // them, out-of-range ones included), so they may stay in flight:
      // everything older -- this stage's LDS-DMA -- is complete.
      if constexpr (WPACK)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SIP_STAMP(ts_b);
      acc_bwait += ts_b - ts_a;
      SIP_SEG(0);
      if (i > 0) { // next stage streams into the other buffer meanwhile
        issue_backward(i - 1, lds + (PAR ^ 1) * C::B_BYTES);
        asm volatile("" ::: "memory"); // stores of this stage stay younger
      }
      lds_cdouble *nm = (lds_cdouble *)(buf + rr * C::BM::ROW_BYTES);
      lds_cdouble *ea = SPLIT ? (lds_cdouble *)(buf + C::BM::BYTES + rr * C::BA::ROW_BYTES) : nm + L::NODE;
      lds_cdouble *nv = (lds_cdouble *)(buf + C::BM::BYTES + C::BA::BYTES + rr * C::BV::ROW_BYTES);
      SIP_SEG(1);
      NodeTail nt;
      backward_edge(i, nm, ea, nv, nt);
      finish_node(i, nt);
    } else {
      const double *nm = pm + (long)i * STG;
      const double *nv = pv + (long)i * VSTG;
      SIP_SEG(1);
      NodeTail nt;
      backward_edge(i, nm, nm + L::NODE, nv, nt);
      finish_node(i, nt);
    }
    SIP_SEG(9);
  };
#ifdef SIP_QW16_NO_PARITY_UNROLL
  constexpr bool kParityUnroll = false;
#else
  constexpr bool kParityUnroll = STAGED;
#endif
  if constexpr (kParityUnroll) {
    // trips over (odd i, i - 1); the first trip of an odd T has no odd stage
    for (int i = (T - 1) | 1; i >= 1; i -= 2) {
      if (i <= T - 1)
        stage_step(i, std::integral_constant<int, 1>{});
      stage_step(i - 1, std::integral_constant<int, 0>{});
    }
  } else if constexpr (STAGED) {
    for (int i = T - 1; i >= 0; --i) {
      if (i & 1)
        stage_step(i, std::integral_constant<int, 1>{});
      else
        stage_step(i, std::integral_constant<int, 0>{});
    }
  } else {
    for (int i = T - 1; i >= 0; --i)
      stage_step(i, std::integral_constant<int, 0>{});
  }

  SIP_STAMP(ts_bwd);
  // ---- root: g_0 = v_0 + W_0 (c_0 - delta_0 o v_0)  (lqr.cpp:798-819) -----
  if constexpr (VDIST) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    dotv<N, true>(acc, td, W);
    if (valid)
      pw[WG + c] = vd + sum4(acc);
  } else {
    double F[N], tt[N];
    sfor<0, N>([&](auto ii) {
      constexpr int r = decltype(ii)::value;
      if constexpr (STAGED) {
        tt[r] = isV ? my_t[r] : 0.0;
        F[r] = isV ? my_v[r] : 0.0;
      } else {
        tt[r] = t[r];
        F[r] = vch[r];
      }
    });
    rank1x<N, N, true>(F, W, tt);
    if (valid && isV) {
      double *gn = pw + WG;
      sfor<0, N>([&](auto ii) { gn[decltype(ii)::value] = F[decltype(ii)::value]; });
    }
  }
  if (valid && c == 0)
    status[p] = stat;
  } else if constexpr (!SPLIT) {
    // ---- mode 2: the affine sweep alone (lqr.cpp:738-796), vectors distributed over the lanes ----
    // Node step: t = c - delta o v; h = S D^{-1/2} t (spilled for the rollout); returns
    // W t = D^{-1/2} (I - S) D^{-1/2} t.
    int woff[N];
    spill_row_offsets(woff);
    auto node_step = [&](const int i, const double v_) {
      const double dl = pm[(long)i * STG + L::QLEN + cm];
      const double t_ = pv[(long)i * VSTG + N + cm] - dl * v_;
      const double sdi = rsqrt_nr(dl);
      const double phi = sdi * t_;
      const double *slot = pw + (long)i * WSN;
      double Srow[N];
      sfor<0, N>([&](auto kk) { Srow[decltype(kk)::value] = slot[woff[decltype(kk)::value]]; });
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, phi, Srow);
      const double sphi = sum4(acc);
      if (valid && isM)
        pw[(long)i * WSN + WG + N + c] = sphi;
      return sdi * (phi - sphi);
    };
    double v_ = pv[(long)T * VSTG + cm]; // v_T = q_T
    double wt = node_step(T, v_);
    for (int i = T - 1; i >= 0; --i) {
      const double *em = pm + (long)i * STG + L::NODE;
      const double *gi = pg + (long)i * L::GAIN;
      const double *gf = gfac + (p * (long)T + i) * (M * M + M);
      const double gd = v_ + wt; // g = v_c + W t  (lqr.cpp:780-781)
      if (valid && isM)
        pw[(long)(i + 1) * WSN + WG + c] = gd;
      double Bcol[N], Acol[N], Kc[M], hf[M], kf[M], Lf[M][M], rinv[M];
      sfor<0, N>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        Bcol[k] = em[N * N + cu * N + k];
        Acol[k] = em[cm * N + k];
      });
      sfor<0, M>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        Kc[j] = gi[cm * M + j];
        rinv[j] = gf[M * M + j];
        sfor<0, j>([&](auto iv) { Lf[j][decltype(iv)::value] = gf[decltype(iv)::value * M + j]; });
      });
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc, gd, Bcol);
      const double hd = pv[(long)i * VSTG + L::VNODE + cu] + sum4(acc); // h = r + B^T g  (:783-784)
      sfor<0, M>([&](auto jj) { hf[decltype(jj)::value] = bcast<decltype(jj)::value>(hd); });
      sfor<0, M>([&](auto jj) { // k = -G^{-1} h  (:785-791), replicated
        constexpr int j = decltype(jj)::value;
        double w = hf[j];
        sfor<0, j>([&](auto iv) { w = __builtin_fma(-Lf[j][decltype(iv)::value], kf[decltype(iv)::value], w); });
        kf[j] = w * rinv[j];
      });
      sfor_down<M - 1, -1>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        double a = 0.0;
        sfor<j + 1, M>([&](auto iv) { a = __builtin_fma(Lf[decltype(iv)::value][j], kf[decltype(iv)::value], a); });
        kf[j] = __builtin_fma(-rinv[j], a, kf[j]);
      });
      if (valid && c == 0) {
        double *gk = pg + (long)i * L::GAIN + N * M;
        sfor<0, M>([&](auto jj) { gk[decltype(jj)::value] = -kf[decltype(jj)::value]; });
      }
      double acc2[4] = {0.0, 0.0, 0.0, 0.0};
      dotv<N, true>(acc2, gd, Acol);
      double kh = 0.0; // K^T h with K = -G^{-1} H (the stored gain)
      sfor<0, M>([&](auto jj) { kh = __builtin_fma(Kc[decltype(jj)::value], hf[decltype(jj)::value], kh); });
      v_ = pv[(long)i * VSTG + cm] + sum4(acc2) + kh; // v = q + A^T g + K^T h  (:793-794)
      wt = node_step(i, v_);
    }
    if (valid && isM)
      pw[WG + c] = v_ + wt; // g_0
  }

  if (mode == 1) // split sip_lqr_factor(): gains, G factors and statuses only, no rollout
    return;
  // The rollout reads W / g / K / k written above by other lanes of this
  // wave: workgroup-scope release/acquire (the block is one wavefront).
  __syncthreads();
  SIP_STAMP(ts_root);

  // ---- forward rollout (lqr.cpp:821-870); lane r < N owns row r ----------

  typename C::FA dma_fa;
  typename C::FG dma_fg;
  typename C::FW dma_fw;
  typename C::FC dma_fd; // delta of the child node
  if constexpr (STAGED) {
    dma_fa.init(lane, (unsigned)((SPLIT ? ab_pstride : mats_len) * 8), max_rel);
    dma_fg.init(lane, (unsigned)(gains_len * 8), max_rel);
    dma_fw.init(lane, (unsigned)(ws_len * 8), max_rel);
    dma_fd.init(lane, (unsigned)(mats_len * 8), max_rel);
  }
  auto issue_forward = [&](const int i, lds_char *buf) {
    dma_fa.template issue<SIP_LQR_NT_FAB>(
        (const char *)(SPLIT ? ab + p0 * ab_pstride + (long)i * ab_sstride : mats + p0 * mats_len + (long)i * STG + L::NODE),
        buf, lane);
    dma_fg.template issue<SIP_LQR_NT_FSP>((const char *)(gains + p0 * gains_len + (long)i * L::GAIN),
                 buf + C::FA::BYTES, lane);
    dma_fw.template issue<SIP_LQR_NT_FSP>((const char *)(wsp + p0 * ws_len + (long)(i + 1) * WSN),
                 buf + C::FA::BYTES + C::FG::BYTES, lane);
    dma_fd.template issue<SIP_LQR_NT_FAB>(
        (const char *)(mats + p0 * mats_len + (long)(i + 1) * STG + L::QLEN),
        buf + C::FA::BYTES + C::FG::BYTES + C::FW::BYTES, lane);
  };
  if constexpr (STAGED) {
    if (T > 0)
      issue_forward(0, lds);
    if (C::F_NBUF == 3 && T > 1)
      issue_forward(1, lds + C::F_BYTES);
  }

  // root (lqr.cpp:798-819): x_0 = -(I + Delta V)^{-1}(delta o v - c) = D^{1/2} h_0,
  // y_0 = v_0 + V_0 x_0 = g_0
  double x, y;
  {
    const double gg = pw[WG + cm];
    const double hh = pw[WG + N + cm];
    const double dd = pm[L::QLEN + cm];
    y = gg;
    x = (dd * rsqrt_nr(dd)) * hh;
    if (valid && isM) {
      ps[c] = x;
      ps[N + c] = y;
    }
  }
  int woff[N];
  spill_row_offsets(woff);
  int fbuf = 0; // LDS buffer of stage i = i % F_NBUF
  for (int i = 0; i < T; ++i) {
    double KT[N], Arow[N], Brow[M], Wc[N];
    double kk0, gg, hh, dd;
    auto read_stage = [&](auto em, auto gi, auto wn, auto dp) {
      sfor<0, N>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        KT[k] = gi[k * M + cu];
        Arow[k] = em[k * N + cm];
        Wc[k] = wn[woff[k]];
      });
      sfor<0, M>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        Brow[j] = em[N * N + j * N + cm];
      });
      kk0 = gi[N * M + cu];
      gg = wn[WG + cm];
      hh = wn[WG + N + cm];
      dd = dp[cm];
    };
    SIP_STAMP(ts_a);
#ifdef SIP_LQR_STAMPS
    if (stamps != nullptr && lane == 0 && i < 64)
      stamps[(long)gridDim.x * 24 + (long)blockIdx.x * 128 + 64 + i] = ts_a;
#endif
    if constexpr (STAGED) {
      lds_char *buf = lds + (fbuf * C::F_BYTES);
      // Stage i+1's LDS-DMA (F_GLDS instructions, the youngest vector-memory
      // operations of this wave: issued after the previous iteration's
      // stores) may stay in flight; everything older -- stage i -- has landed.
      static_assert(C::F_GLDS < 64, "vmcnt is 6 bits");
      if (C::F_NBUF == 3 && i + 1 < T)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::F_GLDS) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SIP_STAMP(ts_b);
      acc_fwait += ts_b - ts_a;
      lds_char *b1 = buf + C::FA::BYTES, *b2 = b1 + C::FG::BYTES,
               *b3 = b2 + C::FW::BYTES;
      read_stage((lds_cdouble *)(buf + rr * C::FA::ROW_BYTES),
                 (lds_cdouble *)(b1 + rr * C::FG::ROW_BYTES),
                 (lds_cdouble *)(b2 + rr * (WSN * 8)),
                 (lds_cdouble *)(b3 + rr * C::FC::ROW_BYTES));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (C::F_NBUF == 2 && i + 1 < T) // one stage ahead, behind this stage's arithmetic
        issue_forward(i + 1, lds + (fbuf ^ 1) * C::F_BYTES);
    } else {
      read_stage(pm + (long)i * STG + L::NODE, pg + (long)i * L::GAIN,
                 pw + (long)(i + 1) * WSN, pm + (long)(i + 1) * STG + L::QLEN);
    }

    const double sdi = rsqrt_nr(dd), sdv = dd * sdi; // as node_factor computed them
    double acc[4];
    // u = k + K x  (lqr.cpp:856-857); lanes < M
    acc[0] = kk0, acc[1] = 0.0, acc[2] = 0.0, acc[3] = 0.0;
    dotv<N, true>(acc, x, KT);
    const double u = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // z = A x + B u  (lqr.cpp:861-862)
    acc[0] = 0.0, acc[1] = 0.0, acc[2] = 0.0, acc[3] = 0.0;
    dotv<N, true>(acc, x, Arow);
    dotv<M, true>(acc, u, Brow);
    const double z = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // zeta = D^{-1/2} z;  x_c = D^{1/2} (S zeta + h);  y_c = g_c + D^{-1/2} (zeta - S zeta)
    const double zeta = z * sdi;
    acc[0] = 0.0, acc[1] = 0.0, acc[2] = 0.0, acc[3] = 0.0;
    dotv<N, true>(acc, zeta, Wc);
    const double sz = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    x = sdv * (sz + hh);
    y = __builtin_fma(sdi, zeta - sz, gg);
    if (valid) {
      double *si = ps + (long)i * VSTG;
#if SIP_LQR_NT_SOL
      if (c < M)
        __builtin_nontemporal_store(u, si + 2 * N + c);
      if (isM) {
        __builtin_nontemporal_store(x, si + VSTG + c);
        __builtin_nontemporal_store(y, si + VSTG + N + c);
      }
#else
      if (c < M)
        si[2 * N + c] = u;
      if (isM) {
        si[VSTG + c] = x;
        si[VSTG + N + c] = y;
      }
#endif
    }
    if constexpr (STAGED) {
      // Keep the stores above older than the DMA below (see the vmcnt count).
      asm volatile("" ::: "memory");
      const int nbuf = fbuf >= 1 ? fbuf - 1 : C::F_NBUF - 1; // (i + 2) % 3
      if (C::F_NBUF == 3 && i + 2 < T)
        issue_forward(i + 2, lds + nbuf * C::F_BYTES);
      fbuf = fbuf + 1 == C::F_NBUF ? 0 : fbuf + 1;
    }
  }
#ifdef SIP_LQR_STAMPS
  SIP_STAMP(ts_end);
  if (stamps != nullptr && lane == 0) {
    unsigned long long *o = stamps + (long)blockIdx.x * 24;
    o[0] = ts_begin, o[1] = ts_term, o[2] = ts_bwd, o[3] = ts_root;
    o[4] = ts_end, o[5] = acc_bwait, o[6] = acc_fwait;
    o[7] = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < 12; ++k)
      o[8 + k] = seg[k];
  }
#endif
}

} // namespace sipamd
