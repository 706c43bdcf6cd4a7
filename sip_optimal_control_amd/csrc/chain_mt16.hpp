// chain_mt16.hpp -- batched chain Riccati kernel for n = 32 on 16 x 16 matrix-core tiles
// (BASELINE config 4: batch 4096, T = 100, n = 32, m = 8, fp32; the same code in fp64).
//
// One problem per wavefront.  A 32 x 32 matrix is 2 x 2 tiles in the C/D layout of
// v_mfma_{f32,f64}_16x16x4: lane (j = lane & 15, g = lane >> 4), register v of tile (I, J) holds
// element (16 I + row(g, v), 16 J + j), row(g, v) = 4 g + v in fp32 and g + 4 v in fp64.  In that layout
//
//      sum_v mfma(U[v], V[v]) = U^T V            (one 16 x 16 tile, contraction over the tile's rows)
//
// takes both operands as they stand (register v of lane (j, g) is A[i = j][k = g] of one operand and
// B[k = g][j] of the other), so no operand of the recursion ever goes through LDS:
//      F = W A = W^T A                         lqr.cpp:703          32 MFMAs
//      Z = W B                                 (H_child^T, :692)     16
//      G = R + B^T Z                           lqr.cpp:693-694        8
//      H = M^T + B^T F                         lqr.cpp:704-705       16
//      K = -G^-1 H                             lqr.cpp:707-713        4
//      V = Q + A^T F + K^T H                   lqr.cpp:715-719       32 + 8
// Against the 32x32x2 formulation of chain_mf32.hpp the m-row blocks (Z, G, H, K: 8 of 32 rows or
// columns) cost a 16-wide tile instead of a 32-wide one.  Controls sit at the tile rows / columns
// row(a & 3, a >> 2): registers 0 and 1 of every lane group, so a contraction over the control index
// (K = Gn^T H, K^T H) issues two MFMAs per tile, not four.
//
// The factorisations are symmetric SWEEPS on the matrix pipe, four pivots per step: with K the pivot
// set, C = A[:, K], P = A[K, K],
//      A[i][j] <- A[i][j] - C_i P^-1 C_j^T  (i, j not in K),  A[:, K] <- C P^-1,  A[K, K] <- -P^-1,
// and after all pivots A holds -A^-1 (the sweep operator).  The pivots met on the way are the Schur
// complements a Cholesky meets (d_k = L_kk^2 of Eigen::LLT, lqr.cpp:505 / :697, in this pivot order),
// so "pivot <= 0" <=> the reference's failure status.  One sweep replaces Cholesky + triangular
// inverse + X^T X of chain_mf32.hpp (48 MFMAs of 32x32x2 -> the equivalent of 24), and W follows as
// W = D^-1/2 (I - F^-1) D^-1/2 (lqr.cpp:521-528) elementwise.  The pivot set of a step is register v of
// the four lane groups (rows row(0..3, v) of tile row I): then C^T is register v of tile row I AS IT
// STANDS for both operands of the update
//      A += (-C + [I on the pivot rows]) * (P^-1 (C^T - [I on the pivot columns])),
// whose modified operands also write the pivot rows and columns (C P^-1) and leave 2 I - P^-1 on the
// pivot block (corrected by a subtraction on four lanes).  P^-1 (4 x 4, from an LDL^T in registers,
// uniform over the wave) enters through one more MFMA per tile column ("mix").
//
// Vectors (g, h, k, v of the affine sweep; x, u, y of the rollout) are small LDS arrays and
// broadcast-FMA sums over the lane groups.
#pragma once
#include <hip/hip_runtime.h>

// Diagnostic build (tools/dev/mt16_stamps.hip): cycles per segment of the backward loop, per wavefront.
#ifdef SIP_MT16_STAMPS
#define SIP_MT16_STAMP_ARG , unsigned long long *__restrict__ stamps
#define SIP_MT16_STAMP(k)                                                                                     \
  do {                                                                                                       \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                            \
    seg_[k] += now_ - last_;                                                                                 \
    last_ = now_;                                                                                            \
  } while (0)
#else
#define SIP_MT16_STAMP_ARG
#define SIP_MT16_STAMP(k)
#endif

namespace sipamd {
namespace mt16 {

constexpr int N = 32;

template <typename S> struct Tr;
template <> struct Tr<float> {
  typedef float v4 __attribute__((ext_vector_type(4)));
  static constexpr bool ROWS_CONTIGUOUS = true; // registers 0..3 of a lane are consecutive rows
  static constexpr int RSTEP = 1;                // row(g, v) = row(g, 0) + RSTEP * v
  static __device__ __forceinline__ constexpr int row(const int g, const int v) { return 4 * g + v; }
  static __device__ __forceinline__ v4 mfma(const float a, const float b, const v4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float rcp(const float x) {
    return __builtin_amdgcn_rcpf(x); // 1 ulp; the pivots' reciprocals enter P^-1 = M^T D^-1 M linearly
  }
  static __device__ __forceinline__ float uniform(const float x, const int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
  }
  static __device__ __forceinline__ float sqrt(const float x) { return sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(const float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float fma(const float a, const float b, const float c) { return __builtin_fmaf(a, b, c); }
};
template <> struct Tr<double> {
  typedef double v4 __attribute__((ext_vector_type(4)));
  static constexpr bool ROWS_CONTIGUOUS = false;
  static constexpr int RSTEP = 4;
  static __device__ __forceinline__ constexpr int row(const int g, const int v) { return g + 4 * v; }
  static __device__ __forceinline__ v4 mfma(const double a, const double b, const v4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ double rcp(const double x) { return 1.0 / x; }
  static __device__ __forceinline__ double uniform(const double x, const int lane) {
    const long long bits = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
  static __device__ __forceinline__ double sqrt(const double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rsqrt(const double x) { return 1.0 / ::sqrt(x); }
  static __device__ __forceinline__ double fma(const double a, const double b, const double c) { return __builtin_fma(a, b, c); }
};

// typed fused multiply-add (__builtin_fma on floats would go through double)
__device__ __forceinline__ float fma_(const float a, const float b, const float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(const double a, const double b, const double c) { return __builtin_fma(a, b, c); }

// The workgroup is ONE wavefront: the LDS instructions of a wavefront execute in order, so a write by some lanes
// is visible to a later read by others without a barrier.  __syncthreads() would also drain vmcnt -- every
// outstanding load, spill store and gains store -- six times per stage; this only keeps the compiler from moving
// memory operations across the point (no instruction is emitted).
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_wave_barrier(); }

// Global memory through buffer instructions: SGPR descriptor (one per array, based at the wavefront's problem), one
// 32-bit VGPR byte offset per lane pattern, an SGPR offset for the stage and a 12-bit immediate -- no 64-bit pointer
// pair per access in the VGPRs.  Offsets in BYTES.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, const long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
template <typename S> struct Mem;
template <> struct Mem<float> {
  typedef Tr<float>::v4 v4;
  static __device__ __forceinline__ float ld(const rsrc_t r, const unsigned vo, const unsigned so) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
  }
  static __device__ __forceinline__ v4 ld4(const rsrc_t r, const unsigned vo, const unsigned so) {
    return __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
  }
  static __device__ __forceinline__ void st(const float x, const rsrc_t r, const unsigned vo, const unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, vo, so, 0);
  }
  static __device__ __forceinline__ void st4(const v4 x, const rsrc_t r, const unsigned vo, const unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), r, vo, so, 0);
  }
};
template <> struct Mem<double> {
  typedef Tr<double>::v4 v4;
  typedef double d2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ double ld(const rsrc_t r, const unsigned vo, const unsigned so) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
  }
  static __device__ __forceinline__ v4 ld4(const rsrc_t r, const unsigned vo, const unsigned so) {
    const d2 a = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    const d2 b = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, vo + 16u, so, 0));
    return v4{a[0], a[1], b[0], b[1]};
  }
  static __device__ __forceinline__ void st(const double x, const rsrc_t r, const unsigned vo, const unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2b, x), r, vo, so, 0);
  }
  static __device__ __forceinline__ void st4(const v4 x, const rsrc_t r, const unsigned vo, const unsigned so) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d2{x[0], x[1]}), r, vo, so, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d2{x[2], x[3]}), r, vo + 16u, so, 0);
  }
};

template <typename S> struct Mat32 { // 32 x 32: t[I][J]
  typename Tr<S>::v4 t[2][2];
};
template <typename S> struct Pair { // 32 x 16 (t[I]: B, Z) or 16 x 32 (t[J]: H, K, M^T)
  typename Tr<S>::v4 t[2];
};

template <typename S> __device__ __forceinline__ typename Tr<S>::v4 zero4() {
  return typename Tr<S>::v4{S(0), S(0), S(0), S(0)};
}

// acc += U^T V (32 x 32 each); four independent accumulators take turns (the 16x16x4 MFMA has a
// 40-cycle dependent latency against a 32-cycle issue interval)
template <typename S>
__device__ __forceinline__ void mul_tt(const Mat32<S> &U, const Mat32<S> &V, Mat32<S> &acc) {
#pragma unroll
  for (int R = 0; R < 2; ++R)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
          acc.t[I][J] = Tr<S>::mfma(U.t[R][I][v], V.t[R][J][v], acc.t[I][J]);
}

// Position of the lane and the control index of a tile row / column.
// Lane selectors of the sweep.  fp64: one-hot vectors in registers (the selection is a handful of FMAs).  fp32,
// where the register budget decides the occupancy: lane MASKS in SGPR pairs, applied by v_cndmask.
typedef unsigned long long lanemask_t;
template <typename S> struct LaneT {
  int lane, j, g;
  S eg[4]; // one-hot of the lane group:      eg[k] = (g == k)
  S ej[4]; // one-hot of the tile rows row(kk, 0): ej[kk] = (j == row(kk, 0))
};
template <> struct LaneT<float> {
  int lane, j, g;
  lanemask_t mg[4], mj[4]; // lanes with g == k / with j == row(kk, 0)
};
// mask[lane] ? a : b.  The masks are laundered through an empty asm when they are built (so that the compiler cannot
// fold them back into comparisons of the lane group, which it then lowers to exec-masked branches -- a switch -- when
// selects on the same value nest); the selects themselves are the compiler's own v_cndmask with an SGPR-pair mask,
// so every hazard around them (transcendental results, MFMA operands) is its business, not an inline asm's.
__device__ __forceinline__ float sel(const lanemask_t m, const float a, const float b) {
  return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b;
}
__device__ __forceinline__ lanemask_t opaque_mask(const bool lane_predicate) {
  lanemask_t m = __builtin_amdgcn_ballot_w64(lane_predicate);
  asm volatile("" : "+s"(m));
  return m;
}

// control index of tile row row(g, v): a = g + 4 v
__device__ __forceinline__ constexpr int ctrl_of(const int g, const int v) { return g + 4 * v; }
// control index of tile COLUMN j (the (g', v') with row(g', v') == j)
template <typename S> __device__ __forceinline__ int ctrl_of_col(const int j) {
  return Tr<S>::ROWS_CONTIGUOUS ? (j >> 2) + 4 * (j & 3) : j;
}

// sum over the four lane groups (lanes j, j + 16, j + 32, j + 48): every lane gets the total.  On the vector
// pipe: v_permlane32_swap / v_permlane16_swap (gfx950) exchange the halves / the odd and even 16-lane rows of two
// registers, so x + swap(x) is the sum with the partner row -- no LDS round trip (ds_bpermute) in the dependent
// chains of the vector work.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sum_groups(const float x) {
  const unsigned b = __float_as_uint(x);
  const u32x2 h = __builtin_amdgcn_permlane32_swap(b, b, false, false); // {[lo, lo], [hi, hi]}
  const float s = __uint_as_float(h[0]) + __uint_as_float(h[1]);
  const unsigned c = __float_as_uint(s);
  const u32x2 q = __builtin_amdgcn_permlane16_swap(c, c, false, false); // rows {[0, 0, 2, 2], [1, 1, 3, 3]}
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ double swap_sum(const double x, const bool rows16) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
  const u32x2 a = rows16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false)
                         : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const u32x2 b = rows16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false)
                         : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double u = __longlong_as_double((long long)(((unsigned long long)b[0] << 32) | a[0]));
  const double w = __longlong_as_double((long long)(((unsigned long long)b[1] << 32) | a[1]));
  return u + w;
}
__device__ __forceinline__ double sum_groups(const double x) { return swap_sum(swap_sum(x, false), true); }

// vec[16 I + row(g, v)], v = 0..3 (LDS array indexed by row)
template <typename S>
__device__ __forceinline__ typename Tr<S>::v4 by_row(const S *vec, const int I, const int g) {
  typename Tr<S>::v4 r;
#pragma unroll
  for (int v = 0; v < 4; ++v)
    r[v] = vec[16 * I + Tr<S>::row(g, v)];
  return r;
}

// y = Mt^T vec: returns the partial sums of this lane group for columns j and 16 + j (sum_groups
// completes them); vec given by row.
template <typename S>
__device__ __forceinline__ void mat_t_vec(const Mat32<S> &Mt, const typename Tr<S>::v4 (&xr)[2], S &p0, S &p1) {
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      p0 = fma_(Mt.t[I][0][v], xr[I][v], p0);
      p1 = fma_(Mt.t[I][1][v], xr[I][v], p1);
    }
}

// ---- the sweep -------------------------------------------------------------------------------
// T: TILES x TILES symmetric matrix in tiles; sweeps the pivots of registers 0 .. VP-1 of every tile
// row (all of them for VP = 4).  Returns true iff a pivot was <= 0.  On return the swept rows and
// columns hold -A^-1 restricted to them (all of -A^-1 for VP = 4).
template <typename S, int TILES, int VP>
__device__ __forceinline__ bool sweep(typename Tr<S>::v4 (&T)[TILES][TILES], const LaneT<S> &L, S *sp) {
  using TR = Tr<S>;
  using v4 = typename TR::v4;
  bool fail = false;
#pragma unroll
  for (int I = 0; I < TILES; ++I) {
#pragma unroll
    for (int v = 0; v < VP; ++v) {
      // P[a][b] = A[p_a][p_b], p_g = 16 I + row(g, v): register v of lane (j = row(b, v), g = a).  Through
      // LDS rather than v_readlane: the 16 lanes that hold P write it compactly, every lane reads it back as
      // four broadcast 16-byte reads -- no vector-pipe time (a v_readlane costs 7 to 26 cycles of it, and the
      // f32 matrix instructions share that pipe), no SGPR operands in the arithmetic below.
      S P[4][4];
      {
        const int b_of_j = TR::ROWS_CONTIGUOUS ? (L.j >> 2) : (L.j & 3); // j = row(b, v') for this b
        const int v_of_j = TR::ROWS_CONTIGUOUS ? (L.j & 3) : (L.j >> 2); //   and this v'
        if (v_of_j == v)
          sp[4 * L.g + b_of_j] = T[I][I][v];
        lds_order();
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b)
            P[a][b] = sp[4 * a + b];
        lds_order();
      }
      // LDL^T of P: the pivots d_k are those of an unblocked Cholesky (squared)
      const S d0 = P[0][0], i0 = TR::rcp(d0);
      const S l10 = P[1][0] * i0, l20 = P[2][0] * i0, l30 = P[3][0] * i0;
      const S d1 = fma_(-l10, P[1][0], P[1][1]), i1 = TR::rcp(d1);
      const S t21 = fma_(-l20, P[1][0], P[2][1]), t31 = fma_(-l30, P[1][0], P[3][1]);
      const S l21 = t21 * i1, l31 = t31 * i1;
      const S d2 = fma_(-l21, t21, fma_(-l20, P[2][0], P[2][2])), i2 = TR::rcp(d2);
      const S t32 = fma_(-l31, t21, fma_(-l30, P[2][0], P[3][2]));
      const S l32 = t32 * i2;
      const S d3 = fma_(-l32, t32, fma_(-l31, t31, fma_(-l30, P[3][0], P[3][3])));
      const S i3 = TR::rcp(d3);
      fail |= !(d0 > S(0)) | !(d1 > S(0)) | !(d2 > S(0)) | !(d3 > S(0));
      // With M = L^-1 (unit lower) and X = C^T - [I on the pivot columns] (register v of tile row I as it stands,
      // minus one on the diagonal lanes) the whole step is  A <- A - Y^T D^-1 Y,  Y = M X :
      //   mix    Y_J = M X_J : the A operand carries M[kk][kq] on lane (i = row(kk, 0), kq = g), so that the result
      //          lands in register 0 of lane group kk -- which is both the B operand layout of Y and the A operand
      //          layout of Y^T;
      //   update A[It][Jt] += (-Y_It)^T-as-A-operand * (d_g^-1 Y_Jt)-as-B-operand.
      // Column g of M for this lane group, by the one-hot of the group (no selects, no P^-1):
      const S m10 = -l10, m21 = -l21, m32 = -l32;
      const S m20 = fma_(l21, l10, -l20);
      const S m31 = fma_(l32, l21, -l31);
      const S m30 = fma_(-l32, m20, fma_(l31, l10, -l30));
      S amix, ig;
      if constexpr (sizeof(S) == 4) { // column g of M = (M e_g): selected by the lane-group masks
        const S e1 = sel(L.mg[1], S(1), S(0)), e2 = sel(L.mg[2], S(1), S(0)), e3 = sel(L.mg[3], S(1), S(0));
        const S t1 = sel(L.mg[0], m10, e1);
        const S t2 = sel(L.mg[0], m20, sel(L.mg[1], m21, e2));
        const S t3 = sel(L.mg[0], m30, sel(L.mg[1], m31, sel(L.mg[2], m32, e3)));
        amix = sel(L.mj[0], sel(L.mg[0], S(1), S(0)), sel(L.mj[1], t1, sel(L.mj[2], t2, sel(L.mj[3], t3, S(0)))));
        ig = sel(L.mg[0], i0, sel(L.mg[1], i1, sel(L.mg[2], i2, i3))); // 1 / d_g
      } else {
        const S t1 = fma_(m10, L.eg[0], L.eg[1]);
        const S t2 = fma_(m20, L.eg[0], fma_(m21, L.eg[1], L.eg[2]));
        const S t3 = fma_(m30, L.eg[0], fma_(m31, L.eg[1], fma_(m32, L.eg[2], L.eg[3])));
        amix = fma_(L.ej[0], L.eg[0], fma_(L.ej[1], t1, fma_(L.ej[2], t2, L.ej[3] * t3)));
        ig = fma_(i0, L.eg[0], fma_(i1, L.eg[1], fma_(i2, L.eg[2], i3 * L.eg[3]))); // 1 / d_g
      }
      const bool diag = L.j == TR::row(L.g, v); // this lane's register v of tile (I, I) is a diagonal element
      const S one_d = diag ? S(1) : S(0);
      S bop[TILES], aop[TILES];
#pragma unroll
      for (int Jt = 0; Jt < TILES; ++Jt) {
        const S src = T[I][Jt][v] - (Jt == I ? one_d : S(0));
        const v4 mixed = TR::mfma(amix, src, zero4<S>());
        aop[Jt] = -mixed[0];
        bop[Jt] = mixed[0] * ig;
      }
#pragma unroll
      for (int It = 0; It < TILES; ++It)
#pragma unroll
        for (int Jt = 0; Jt < TILES; ++Jt)
          T[It][Jt] = TR::mfma(aop[It], bop[Jt], T[It][Jt]);
      T[I][I][v] -= S(2) * one_d;
    }
  }
  return fail;
}

template <typename S, int M> struct Layout {
  static constexpr int NODE = N * N + N;                 // Q | delta
  static constexpr int EDGE = N * N + 2 * N * M + M * M; // A | B | M | R
  static constexpr int VNODE = 2 * N, VEDGE = M, GAIN = M * N + M;
  // spill per node: tiles (0, 0), (1, 0), (1, 1) of the symmetric W as they sit in the registers | g
  static constexpr int WTILES = 3, WSN = WTILES * 256 + N;
  static constexpr int VM = (M + 3) / 4; // registers of a lane that hold controls
  // 16-byte loads of four consecutive rows need every block 16-byte aligned
  static constexpr bool VEC_LOADS = Tr<S>::ROWS_CONTIGUOUS && (EDGE % 4 == 0);
};

// Column-major 32 x 32 (ld 32) at element offset `so` of the array behind `r` -> tiles.  ocol = j * 32 + row(g, 0).
template <typename S, bool VEC>
__device__ __forceinline__ Mat32<S> load32(const rsrc_t r, const unsigned ocol, const unsigned so) {
  constexpr unsigned ES = sizeof(S);
  Mat32<S> m;
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J) {
      if constexpr (VEC) {
        m.t[I][J] = Mem<S>::ld4(r, (ocol + (unsigned)(16 * J * N + 16 * I)) * ES, so * ES);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          m.t[I][J][v] = Mem<S>::ld(r, (ocol + (unsigned)(16 * J * N + 16 * I + Tr<S>::RSTEP * v)) * ES, so * ES);
      }
    }
  return m;
}

// Waves per SIMD the register allocator is held to.  fp32: 4 (128 VGPRs) -- BASELINE's batch of 4096 is exactly four
// wavefronts per SIMD, and with three resident the fourth runs alone behind them at a lone wavefront's latency.  What
// made 128 registers enough (the first version spilled ~30 loop-invariant lane constants there and was slower than
// three wavefronts): every global access through a buffer descriptor in SGPRs + one 32-bit lane offset + an SGPR
// stage offset + an immediate (no 64-bit pointer pair per access pattern), and the sweep's lane selectors as SGPR
// masks instead of one-hot vectors.  bench.py --workload c4: 2.43 ms at 3 wavefronts with pointer addressing, 2.37
// with buffer addressing, 2.22 at 4 (7 dwords of scratch left).
#ifndef SIP_MT16_WAVES_F32
#define SIP_MT16_WAVES_F32 4
#endif
template <typename S> struct Waves { static constexpr int value = sizeof(S) == 4 ? SIP_MT16_WAVES_F32 : 2; };

template <typename S, int M>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(Waves<S>::value, Waves<S>::value)))
void chain_factor_solve_mt16(
    const S *__restrict__ mats, const S *__restrict__ vecs, S *__restrict__ sol, S *__restrict__ gains,
    S *__restrict__ wsp, int *__restrict__ status, const long batch, const int T SIP_MT16_STAMP_ARG) {
  static_assert(M >= 1 && M <= 8, "controls live in registers 0 and 1 of the four lane groups");
#ifdef SIP_MT16_STAMPS
  unsigned long long seg_[16] = {0}, last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long start_ = last_;
#endif
  using TR = Tr<S>;
  using v4 = typename TR::v4;
  using LY = Layout<S, M>;
  constexpr int STG = LY::NODE + LY::EDGE, VSTG = LY::VNODE + LY::VEDGE, VM = LY::VM;
  constexpr bool VEC = LY::VEC_LOADS;
  const long p = blockIdx.x;
  if (p >= batch)
    return;
  LaneT<S> L;
  L.lane = threadIdx.x & 63, L.j = L.lane & 15, L.g = L.lane >> 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if constexpr (sizeof(S) == 4) {
      L.mg[k] = opaque_mask(L.g == k);
      L.mj[k] = opaque_mask(L.j == TR::row(k, 0));
    } else {
      L.eg[k] = L.g == k ? S(1) : S(0);
      L.ej[k] = L.j == TR::row(k, 0) ? S(1) : S(0);
    }
  }
  const int j = L.j, g = L.g;
  const int acol = ctrl_of_col<S>(j); // control index of this lane's tile column (valid if < M)
  constexpr unsigned ES = sizeof(S);
  const long mats_len = (long)(T + 1) * LY::NODE + (long)T * LY::EDGE, vecs_len = (long)(T + 1) * LY::VNODE + (long)T * LY::VEDGE;
  // one buffer descriptor per array, based at this wavefront's problem (SGPRs); every access below is
  // descriptor + 32-bit lane offset + stage offset (SGPR) + immediate
  const rsrc_t rM = make_rsrc(mats + p * mats_len, mats_len * ES), rV = make_rsrc(vecs + p * vecs_len, vecs_len * ES);
  const rsrc_t rS = make_rsrc(sol + p * vecs_len, vecs_len * ES);
  const rsrc_t rG = make_rsrc(gains + p * ((long)T * LY::GAIN), (long)T * LY::GAIN * ES);
  const rsrc_t rW = make_rsrc(wsp + p * ((long)(T + 1) * LY::WSN), (long)(T + 1) * LY::WSN * ES);
  const unsigned acolc = (unsigned)(acol < M ? acol : 0);
  const unsigned uj = (unsigned)j, ug = (unsigned)g, r0 = (unsigned)TR::row(g, 0);
  const unsigned ocol = uj * N + r0;     // column-major tile: element (row(g, 0), j)
  const unsigned obcol = acolc * N + r0; // ... of B: the column of the control of tile column j
  const unsigned orow = r0 * N + uj;     // transposed tile (A^T): element (j, row(g, 0)) of A
  const unsigned octl = ug * N + uj;     // control rows: element (j, g) of B or M; control g + 4 v is 4 v N further
  constexpr unsigned RS = (unsigned)TR::RSTEP;
  auto ldM = [&](const unsigned vo, const unsigned so) { return Mem<S>::ld(rM, vo * ES, so * ES); };
  auto ldV = [&](const unsigned vo, const unsigned so) { return Mem<S>::ld(rV, vo * ES, so * ES); };

  __shared__ S s_v[N], s_t[N], s_g[N], s_sd[N], s_sdi[N], s_x[N], s_z[N], s_gs[16], s_h[8], s_u[8];
  constexpr int LDM = TR::ROWS_CONTIGUOUS ? 36 : 33; // column stride of the mirror image (bank spread; 16-byte columns)
  __shared__ __attribute__((aligned(16))) S s_m[N * LDM];
  __shared__ __attribute__((aligned(16))) S s_p[16]; // pivot block of the sweep's current step


  if (L.lane < 8)
    s_h[L.lane] = S(0), s_u[L.lane] = S(0); // the entries of no control are read as zeros
  int stat = 0;
  Mat32<S> W, V;

  // ---- node tail: F = I + D^1/2 V D^1/2, its sweep, W (lqr.cpp:487-529, 722-727), plus the vector
  // term t = c - delta o v for the parent step.
  auto finish_node = [&](const int i, const S dl0, const S dl1, const S c0, const S c1) {
    if (stat == 0 && __any(!(dl0 > S(0)) || !(dl1 > S(0))))
      stat = 1; // INVALID_DELTA
    const S sd0 = TR::sqrt(dl0), sd1 = TR::sqrt(dl1);
    const S sdi0 = S(1) / sd0, sdi1 = S(1) / sd1;
    if (g == 0) {
      s_sd[j] = sd0, s_sd[16 + j] = sd1;
      s_sdi[j] = sdi0, s_sdi[16 + j] = sdi1;
      s_t[j] = c0 - dl0 * s_v[j]; // c - delta o v   (lqr.cpp:778-779, negated)
      s_t[16 + j] = c1 - dl1 * s_v[16 + j];
    }
    SIP_MT16_STAMP(7);
    // Only the lower triangle of V counts (Eigen::LLT reads nothing else, lqr.cpp:505): mirror it through
    // LDS.  V = Q + A^T F + K^T H is symmetric only up to rounding, the sweep reads rows as columns, and the
    // recursion does not damp an antisymmetric part (the feedback term K^T H is symmetric by construction):
    // left alone it grows by |A|^2 per stage.
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) { // tile (0, 1) is not read back
        S *col = s_m + (16 * J + j) * LDM + 16 * I;
        if constexpr (TR::ROWS_CONTIGUOUS) {
          *(v4 *)(col + 4 * g) = V.t[I][J];
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v)
            col[TR::row(g, v)] = V.t[I][J][v];
        }
      }
    lds_order();
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = I; J < 2; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * I + TR::row(g, v), c = 16 * J + j;
          const S mirrored = s_m[r * LDM + c]; // element (c, r)
          V.t[I][J][v] = (J > I || r < c) ? mirrored : V.t[I][J][v];
        }
    // F = I + sd V sd (lqr.cpp:497-503), in place
    {
      const v4 sr[2] = {by_row<S>(s_sd, 0, g), by_row<S>(s_sd, 1, g)};
#pragma unroll
      for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const bool dg = I == J && j == TR::row(g, v);
            V.t[I][J][v] = fma_(V.t[I][J][v] * sr[I][v], J == 0 ? sd0 : sd1, dg ? S(1) : S(0));
          }
    }
    SIP_MT16_STAMP(8);
    const bool ffail = sweep<S, 2, 4>(V.t, L, s_p); // V now holds -F^-1
    SIP_MT16_STAMP(9);
    if (stat == 0 && ffail)
      stat = 2; // F_FACTORIZATION_FAILURE
    // W = D^-1/2 (I - F^-1) D^-1/2
    {
      const v4 ir[2] = {by_row<S>(s_sdi, 0, g), by_row<S>(s_sdi, 1, g)};
#pragma unroll
      for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const bool dg = I == J && j == TR::row(g, v);
            W.t[I][J][v] = (V.t[I][J][v] + (dg ? S(1) : S(0))) * (ir[I][v] * (J == 0 ? sdi0 : sdi1));
          }
    }
    // spill W for the rollout: W is symmetric, tile (0, 1) is read back out of the dump of tile (1, 0)
    {
      const unsigned so = (unsigned)i * LY::WSN * ES, vo = (unsigned)L.lane * 4u * ES;
      Mem<S>::st4(W.t[0][0], rW, vo, so);
      Mem<S>::st4(W.t[1][0], rW, vo + 256u * ES, so);
      Mem<S>::st4(W.t[1][1], rW, vo + 512u * ES, so);
    }
    lds_order();
    SIP_MT16_STAMP(10);
  };

  // ---- terminal node -------------------------------------------------------------------------
  {
    const unsigned sM = (unsigned)T * STG, sV = (unsigned)T * VSTG;
    V = load32<S, VEC>(rM, ocol, sM);
    const S qT0 = ldV(uj, sV), qT1 = ldV(uj + 16u, sV);
    if (g == 0)
      s_v[j] = qT0, s_v[16 + j] = qT1; // v = q
    const S dl0 = ldM(uj + (unsigned)(N * N), sM), dl1 = ldM(uj + (unsigned)(N * N + 16), sM);
    const S c0 = ldV(uj + (unsigned)N, sV), c1 = ldV(uj + (unsigned)(N + 16), sV);
    lds_order();
    finish_node(T, dl0, dl1, c0, c1);
  }

  // ---- backward recursion --------------------------------------------------------------------
  for (int i = T - 1; i >= 0; --i) {
    // element offsets of the stage's blocks inside the problem (SGPRs)
    const unsigned sN = (unsigned)i * STG, sE = sN + LY::NODE, sB = sE + N * N, sMm = sB + N * M, sR = sMm + N * M;
    const unsigned sV = (unsigned)i * VSTG;
    const Mat32<S> A = load32<S, VEC>(rM, ocol, sE);
    // the node's and the edge's vectors: requested here, used at the bottom of the stage
    const S dl0 = ldM(uj + (unsigned)(N * N), sN), dl1 = ldM(uj + (unsigned)(N * N + 16), sN);
    const S c0 = ldV(uj + (unsigned)N, sV), c1 = ldV(uj + (unsigned)(N + 16), sV);
    const S q0 = ldV(uj, sV), q1 = ldV(uj + 16u, sV), rv = ldV(acolc + (unsigned)LY::VNODE, sV);
    // B (32 x M, ld 32): control acol on tile column j; columns of no control are zero
    Pair<S> B;
#pragma unroll
    for (int I = 0; I < 2; ++I) {
      if constexpr (VEC) {
        B.t[I] = Mem<S>::ld4(rM, (obcol + (unsigned)(16 * I)) * ES, sB * ES);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          B.t[I][v] = ldM(obcol + (unsigned)(16 * I) + RS * v, sB);
      }
      if (acol >= M)
        B.t[I] = zero4<S>();
    }
    // H starts as M^T (row = control ctrl_of(g, v), column 16 J + j), G as R with an identity on the
    // rows / columns of no control
    Pair<S> H;
    v4 G;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int a = ctrl_of(g, v);
      const bool ok = v < VM && a < M;
      if (v < VM) { // unconditional loads (a row a >= M of the 8-row window lies in the blocks behind: read, dropped)
        const S h0 = ldM(octl + (unsigned)(4 * v * N), sMm), h1 = ldM(octl + (unsigned)(4 * v * N + 16), sMm);
        const S rr = ldM(acolc * M + ug + (unsigned)(4 * v), sR);
        H.t[0][v] = ok ? h0 : S(0);
        H.t[1][v] = ok ? h1 : S(0);
        G[v] = (ok && acol < M) ? rr : (j == TR::row(g, v) ? S(1) : S(0));
      } else {
        H.t[0][v] = S(0), H.t[1][v] = S(0);
        G[v] = j == TR::row(g, v) ? S(1) : S(0);
      }
    }
    // g = v_c + W t   (lqr.cpp:778-781)
    {
      const v4 tr[2] = {by_row<S>(s_t, 0, g), by_row<S>(s_t, 1, g)};
      S w0 = S(0), w1 = S(0);
      mat_t_vec<S>(W, tr, w0, w1);
      const S g0 = s_v[j] + sum_groups(w0), g1 = s_v[16 + j] + sum_groups(w1);
      if (g == 0) {
        s_g[j] = g0, s_g[16 + j] = g1;
        const unsigned so = ((unsigned)(i + 1) * LY::WSN + LY::WTILES * 256) * ES;
        Mem<S>::st(g0, rW, uj * ES, so);
        Mem<S>::st(g1, rW, (uj + 16u) * ES, so);
      }
    }
    SIP_MT16_STAMP(0);
    Mat32<S> F;
    Pair<S> Z;
#pragma unroll
    for (int I = 0; I < 2; ++I) {
      Z.t[I] = zero4<S>();
#pragma unroll
      for (int J = 0; J < 2; ++J)
        F.t[I][J] = zero4<S>();
    }
    // Z = W B and F = W A, interleaved: six independent accumulators
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int I = 0; I < 2; ++I) {
          Z.t[I] = TR::mfma(W.t[R][I][v], B.t[R][v], Z.t[I]);
#pragma unroll
          for (int J = 0; J < 2; ++J)
            F.t[I][J] = TR::mfma(W.t[R][I][v], A.t[R][J][v], F.t[I][J]);
        }
    // G = R + B^T Z, H = M^T + B^T F
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        G = TR::mfma(B.t[R][v], Z.t[R][v], G);
        H.t[0] = TR::mfma(B.t[R][v], F.t[R][0][v], H.t[0]);
        H.t[1] = TR::mfma(B.t[R][v], F.t[R][1][v], H.t[1]);
      }
    SIP_MT16_STAMP(1);
    lds_order(); // s_g
    const v4 gr[2] = {by_row<S>(s_g, 0, g), by_row<S>(s_g, 1, g)};
    // h = r + B^T g  (lqr.cpp:783-784): control acol on the lanes of tile column j
    {
      S pb = S(0);
#pragma unroll
      for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          pb = fma_(B.t[I][v], gr[I][v], pb);
      pb = sum_groups(pb);
      if (g == 0 && acol < M)
        s_h[acol] = rv + pb;
    }
    SIP_MT16_STAMP(2);
    // LLT of G (lqr.cpp:696-701) as a sweep: G <- -G^-1 on the control rows / columns.  The sweep's modified
    // operands work against an identity on the pivot block, which costs accuracy when the pivots are far from
    // 1 (G ~ R is not scaled like F = I + ...): sweep the unit-diagonal S G S, S = diag(G)^-1/2, and scale back.
    {
      S dsel = S(0);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        dsel += j == TR::row(g, v) ? G[v] : S(0);
      const S sc = TR::rsqrt(sum_groups(dsel)); // of tile column j; NaN for a diagonal <= 0: the sweep then fails
      if (g == 0)
        s_gs[j] = sc;
      lds_order();
      const v4 sr = by_row<S>(s_gs, 0, g);
      v4 Gt[1][1];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Gt[0][0][v] = G[v] * (sr[v] * sc);
      const bool gfail = sweep<S, 1, VM>(Gt, L, s_p);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        G[v] = Gt[0][0][v] * (sr[v] * sc);
      if (stat == 0 && gfail)
        stat = 3; // G_FACTORIZATION_FAILURE
    }
    SIP_MT16_STAMP(3);
    // K = -G^-1 H   (lqr.cpp:707-713); -G^-1 symmetric
    Pair<S> K;
    K.t[0] = zero4<S>(), K.t[1] = zero4<S>();
#pragma unroll
    for (int v = 0; v < VM; ++v) {
      K.t[0] = TR::mfma(G[v], H.t[0][v], K.t[0]);
      K.t[1] = TR::mfma(G[v], H.t[1][v], K.t[1]);
    }
    lds_order(); // s_h
    S hr[VM];
#pragma unroll
    for (int v = 0; v < VM; ++v)
      hr[v] = s_h[ctrl_of(g, v)]; // entries >= M stay zero
    // k = -G^-1 h   (lqr.cpp:785-791), control acol on tile column j
    S kb = S(0);
#pragma unroll
    for (int v = 0; v < VM; ++v)
      kb = fma_(G[v], hr[v], kb);
    kb = sum_groups(kb);
    // gains out: K (m x 32 column-major) | k
    {
      const unsigned so = (unsigned)i * LY::GAIN * ES;
#pragma unroll
      for (int v = 0; v < VM; ++v)
        if (ctrl_of(g, v) < M) {
          Mem<S>::st(K.t[0][v], rG, (uj * M + ug + (unsigned)(4 * v)) * ES, so);
          Mem<S>::st(K.t[1][v], rG, (uj * M + ug + (unsigned)(4 * v + 16 * M)) * ES, so);
        }
      if (g == 0 && acol < M)
        Mem<S>::st(kb, rG, (acolc + (unsigned)(M * N)) * ES, so);
    }
    // v = q + A^T g + K^T h   (lqr.cpp:793-794)
    {
      S p0 = S(0), p1 = S(0);
      mat_t_vec<S>(A, gr, p0, p1);
#pragma unroll
      for (int v = 0; v < VM; ++v) {
        p0 = fma_(K.t[0][v], hr[v], p0);
        p1 = fma_(K.t[1][v], hr[v], p1);
      }
      const S vn0 = q0 + sum_groups(p0), vn1 = q1 + sum_groups(p1);
      lds_order(); // everyone is done with s_v / s_t / s_g of the child
      if (g == 0)
        s_v[j] = vn0, s_v[16 + j] = vn1;
    }
    SIP_MT16_STAMP(4);
    // V = Q + A^T F + K^T H   (lqr.cpp:715-719): tiles (0, 0), (1, 0), (1, 1) only -- the node tail keeps the lower
    // triangle and mirrors it, so tile (0, 1) is never read (10 MFMAs and a 16-byte load less per stage)
    {
      const Mat32<S> Qm = load32<S, VEC>(rM, ocol, sN); // (its tile (0, 1) is dead code)
      V.t[0][0] = Qm.t[0][0], V.t[1][0] = Qm.t[1][0], V.t[1][1] = Qm.t[1][1];
    }
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        V.t[0][0] = TR::mfma(A.t[R][0][v], F.t[R][0][v], V.t[0][0]);
        V.t[1][0] = TR::mfma(A.t[R][1][v], F.t[R][0][v], V.t[1][0]);
        V.t[1][1] = TR::mfma(A.t[R][1][v], F.t[R][1][v], V.t[1][1]);
      }
#pragma unroll
    for (int v = 0; v < VM; ++v) {
      V.t[0][0] = TR::mfma(K.t[0][v], H.t[0][v], V.t[0][0]);
      V.t[1][0] = TR::mfma(K.t[1][v], H.t[0][v], V.t[1][0]);
      V.t[1][1] = TR::mfma(K.t[1][v], H.t[1][v], V.t[1][1]);
    }
    lds_order();
    SIP_MT16_STAMP(5);
    finish_node(i, dl0, dl1, c0, c1);
  }
#ifdef SIP_MT16_STAMPS
  const unsigned long long back_end_ = __builtin_amdgcn_s_memtime();
#endif

  // ---- root: g_0 = v_0 + W_0 t_0 ; x_0 = c_0 - delta_0 o g_0, y_0 = g_0 -----------------------
  {
    const v4 tr[2] = {by_row<S>(s_t, 0, g), by_row<S>(s_t, 1, g)};
    S w0 = S(0), w1 = S(0);
    mat_t_vec<S>(W, tr, w0, w1);
    const S g0 = s_v[j] + sum_groups(w0), g1 = s_v[16 + j] + sum_groups(w1);
    if (g == 0) {
      const S x0 = ldV(uj + (unsigned)N, 0u) - ldM(uj + (unsigned)(N * N), 0u) * g0;
      const S x1 = ldV(uj + (unsigned)(N + 16), 0u) - ldM(uj + (unsigned)(N * N + 16), 0u) * g1;
      Mem<S>::st(x0, rS, uj * ES, 0u), Mem<S>::st(x1, rS, (uj + 16u) * ES, 0u);
      Mem<S>::st(g0, rS, (uj + (unsigned)N) * ES, 0u), Mem<S>::st(g1, rS, (uj + (unsigned)(N + 16)) * ES, 0u);
      s_x[j] = x0, s_x[16 + j] = x1;
    }
  }
  if (L.lane == 0)
    status[p] = stat;
  lds_order();

  // ---- forward rollout (lqr.cpp:821-870) -----------------------------------------------------
  // The products sum over the rows of the tile, so the operands are loaded transposed (K^T, A^T, B^T:
  // rows = the index summed over); W is symmetric.  Nothing a stage reads from memory depends on x:
  // every operand of stage i + 1 is requested as soon as stage i has used the registers it lands in
  // (one register set, a stage of lead time).
  Pair<S> KT;   // t[I]: K^T rows 16 I + row(g, v), control acol
  Mat32<S> AT;  // A^T
  Pair<S> BT;   // t[J]: B^T rows = controls (registers < VM), columns 16 J + j
  Mat32<S> Wc;  // W of the child
  S kb, gc0, gc1, cc0, cc1, dc0, dc1;
  auto fetch_K = [&](const int i) {
    const unsigned so = (unsigned)i * LY::GAIN * ES;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int v = 0; v < 4; ++v) // K(acol, r), r = 16 I + row(g, v); lanes of no control: unused
        KT.t[I][v] = Mem<S>::ld(rG, (r0 * M + acolc + (unsigned)((16 * I + TR::RSTEP * v) * M)) * ES, so);
    kb = Mem<S>::ld(rG, (acolc + (unsigned)(M * N)) * ES, so);
  };
  auto fetch_AB = [&](const int i) {
    const unsigned sE = (unsigned)i * STG + LY::NODE;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int v = 0; v < 4; ++v) { // A(j, r), r = 16 I + row(g, v)
        AT.t[I][0][v] = ldM(orow + (unsigned)((16 * I + TR::RSTEP * v) * N), sE);
        AT.t[I][1][v] = ldM(orow + (unsigned)((16 * I + TR::RSTEP * v) * N + 16), sE);
      }
#pragma unroll
    for (int v = 0; v < VM; ++v) {
      const int a = ctrl_of(g, v);
      // B(j, a), a = g + 4 v (a row a >= M of the window lies in the blocks behind B: read, then dropped)
      const S b0 = ldM(octl + (unsigned)(N * N + 4 * v * N), sE), b1 = ldM(octl + (unsigned)(N * N + 4 * v * N + 16), sE);
      BT.t[0][v] = a < M ? b0 : S(0);
      BT.t[1][v] = a < M ? b1 : S(0);
    }
  };
  auto fetch_W = [&](const int i) { // the child's W, g and the node vectors of the child
    const unsigned sN1 = (unsigned)(i + 1) * STG, sV1 = (unsigned)(i + 1) * VSTG, sW = (unsigned)(i + 1) * LY::WSN;
    const unsigned vl = (unsigned)L.lane * 4u * ES;
    Wc.t[0][0] = Mem<S>::ld4(rW, vl, sW * ES), Wc.t[1][0] = Mem<S>::ld4(rW, vl + 256u * ES, sW * ES);
    Wc.t[1][1] = Mem<S>::ld4(rW, vl + 512u * ES, sW * ES);
    // tile (0, 1) = tile (1, 0) transposed: element (row(g, v), 16 + j) is W(16 + j, row(g, v)), which the dump of
    // tile (1, 0) holds in register v' of lane (row(g, v), g') with row(g', v') = j
    {
      const unsigned gq = TR::ROWS_CONTIGUOUS ? (uj >> 2) : (uj & 3u), vq = TR::ROWS_CONTIGUOUS ? (uj & 3u) : (uj >> 2);
      const unsigned ot = (16u * gq + r0) * 4u + vq;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Wc.t[0][1][v] = Mem<S>::ld(rW, (ot + (unsigned)((64 + TR::RSTEP * v) * 4)) * ES, sW * ES);
    }
    gc0 = Mem<S>::ld(rW, (uj + (unsigned)(LY::WTILES * 256)) * ES, sW * ES);
    gc1 = Mem<S>::ld(rW, (uj + (unsigned)(LY::WTILES * 256 + 16)) * ES, sW * ES);
    cc0 = ldV(uj + (unsigned)N, sV1), cc1 = ldV(uj + (unsigned)(N + 16), sV1);
    dc0 = ldM(uj + (unsigned)(N * N), sN1), dc1 = ldM(uj + (unsigned)(N * N + 16), sN1);
  };
  if (T > 0) {
    fetch_K(0);
    fetch_AB(0);
    fetch_W(0);
  }
  for (int i = 0; i < T; ++i) {
    const bool more = i + 1 < T;
    const v4 xr[2] = {by_row<S>(s_x, 0, g), by_row<S>(s_x, 1, g)};
    // u = k + K x : control acol on tile column j
    S pu = S(0);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        pu = fma_(KT.t[I][v], xr[I][v], pu);
    const S ub = kb + sum_groups(pu);
    if (more)
      fetch_K(i + 1);
    if (g == 0 && acol < M)
      s_u[acol] = ub;
    lds_order();
    // z = A x + B u
    S z0 = S(0), z1 = S(0);
    mat_t_vec<S>(AT, xr, z0, z1);
#pragma unroll
    for (int v = 0; v < VM; ++v) {
      const S ur = s_u[ctrl_of(g, v)]; // entries >= M stay zero
      z0 = fma_(BT.t[0][v], ur, z0);
      z1 = fma_(BT.t[1][v], ur, z1);
    }
    if (more)
      fetch_AB(i + 1);
    z0 = sum_groups(z0), z1 = sum_groups(z1);
    if (g == 0)
      s_z[j] = z0, s_z[16 + j] = z1;
    lds_order();
    // y_c = g_c + W_c z ; x_c = z + c_c - delta_c o y_c
    const v4 zr[2] = {by_row<S>(s_z, 0, g), by_row<S>(s_z, 1, g)};
    S y0 = S(0), y1 = S(0);
    mat_t_vec<S>(Wc, zr, y0, y1);
    y0 = gc0 + sum_groups(y0), y1 = gc1 + sum_groups(y1);
    const S xn0 = z0 + (cc0 - dc0 * y0), xn1 = z1 + (cc1 - dc1 * y1);
    if (more)
      fetch_W(i + 1);
    if (g == 0) {
      const unsigned so = (unsigned)i * VSTG * ES;
      if (acol < M)
        Mem<S>::st(ub, rS, (acolc + (unsigned)(2 * N)) * ES, so);
      Mem<S>::st(xn0, rS, (uj + (unsigned)VSTG) * ES, so), Mem<S>::st(xn1, rS, (uj + (unsigned)(VSTG + 16)) * ES, so);
      Mem<S>::st(y0, rS, (uj + (unsigned)(VSTG + N)) * ES, so), Mem<S>::st(y1, rS, (uj + (unsigned)(VSTG + N + 16)) * ES, so);
      s_x[j] = xn0, s_x[16 + j] = xn1;
    }
    lds_order();
  }
#ifdef SIP_MT16_STAMPS
  if (stamps != nullptr && L.lane == 0) {
    unsigned long long *o = stamps + (long)blockIdx.x * 20;
    o[0] = start_, o[1] = back_end_, o[2] = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < 16; ++k)
      o[3 + k] = seg_[k];
  }
#endif
}

} // namespace mt16
} // namespace sipamd
