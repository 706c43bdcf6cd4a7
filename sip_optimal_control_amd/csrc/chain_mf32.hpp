// chain_mf32.hpp -- fp32 batched chain Riccati kernel for n = 32 on the matrix
// cores (BASELINE config 4: batch 4096, T = 100, n = 32, m = 8).
//
// One problem per wavefront.  Every 32 x 32 (and m x 32) matrix lives in
// registers in the C/D layout of v_mfma_f32_32x32x2_f32 -- lane (j, h), reg p
// holds element (row r(p,h) = (p & 3) + 8 (p >> 2) + 4 h, column j) -- because
// in that layout the matrix pipe computes transposed products for free:
//
//      sum_p  mfma_32x32x2( U[p], V[p] )  =  U^T V          (again in C/D layout)
//
// (each MFMA consumes the row pair r(p,0), r(p,1) of both factors as its K = 2
// slice).  Every product of the backward recursion is of that shape:
//      F = W A   = W^T A  (W symmetric)        lqr.cpp:703
//      Z = W B   = W^T B                       (H_child^T of lqr.cpp:692)
//      G = R + B^T Z                           lqr.cpp:693-694
//      H = M^T + B^T F                         lqr.cpp:704-705
//      V = Q + A^T F + K^T H                   lqr.cpp:715-719
//      F^-1 = X^T X with X = L^-1              (the two solves of lqr.cpp:516-519)
// so no operand ever goes through LDS.  The Cholesky of F and the triangular
// inverse run on the matrix pipe too, two columns per step, as rank-2
// eliminations (see finish_node).  Vectors (g, h, k, v of the affine sweep,
// x, u, y of the rollout) are small LDS arrays.
//
// Arithmetic is IEEE fp32 (MFMA f32: exact fmaf chains); parity with the fp64
// oracle is judged by the KKT residual (tests/test_gpu_general_chain.py).
#pragma once
#include <hip/hip_runtime.h>

namespace sipamd {
namespace mf32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 32;

__device__ __forceinline__ int crow(const int p, const int h) {
  return (p & 3) + 8 * (p >> 2) + 4 * h;
}

// acc += U^T V  (all three in C/D layout); ROWS: rows of U / V that can be
// nonzero (32, or 8 for the m-row factors, whose rows live in regs 0..3).
template <int ROWS>
__device__ __forceinline__ void prodT(const f32x16 &U, const f32x16 &V, f32x16 &acc) {
  constexpr int P = ROWS == 32 ? 16 : 4;
#pragma unroll
  for (int p = 0; p < P; ++p)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(U[p], V[p], acc, 0, 0, 0);
}

// Column-major (ld = rows) global matrix -> C/D layout.  Columns >= cols read
// as zero.  32-row matrices: four 16-byte loads per lane.
__device__ __forceinline__ f32x16 load_c32(const float *m, const int cols, const int j, const int h) {
  f32x16 r;
  const f32x4 *src = (const f32x4 *)(m + (j < cols ? j : 0) * N + 4 * h);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = src[2 * g];
    if (j >= cols)
      v = f32x4{0.f, 0.f, 0.f, 0.f};
    r[4 * g + 0] = v[0], r[4 * g + 1] = v[1], r[4 * g + 2] = v[2], r[4 * g + 3] = v[3];
  }
  return r;
}

// y_j = sum_r Mc[r][j] * vec[r]  for the column j of this lane (both halves get
// the full sum).  vec: LDS array indexed by row.
__device__ __forceinline__ float matTvec(const f32x16 &Mc, const float *vec, const int h) {
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < 16; ++p)
    s = __builtin_fmaf(Mc[p], vec[crow(p, h)], s);
  return s + __shfl_xor(s, 32);
}

// Cholesky (Eigen::LLT, lqr.cpp:505 / :697) and the triangular inverse together,
// two columns per step, as rank-2 eliminations on the matrix pipe: with
// [l_k l_k1] the two new columns of L (one entry per lane, by row),
//     A <- A - l_k l_k^T - l_k1 l_k1^T            (trailing update)
//     X <- X - l_k y_k^T - l_k1 y_k1^T, rows > k1   (forward substitution
//                                                  on the identity)
// where y_k, y_k1 are the finished rows k, k1 of X = L^-1.  Row k of the
// symmetric A (= its column k) sits across the lanes of half h_k in reg p_k.
// Pivots are those of the unblocked Cholesky (a_kk - |L_k,0:k|^2), so
// "pivot <= 0" is detected as Eigen does.  PANELS = order / 2; on entry X = I.
// Returns true iff a pivot was <= 0.
template <int PANELS>
__device__ __forceinline__ bool eliminate(f32x16 &Acc, f32x16 &Xc, const int j, const int h) {
  bool fail = false;
#pragma unroll
  for (int pp = 0; pp < PANELS; ++pp) {
    const int k = 2 * pp, k1 = k + 1;
    const int hk = (k >> 2) & 1, pk = (k & 3) + 4 * (k >> 3);
    const float r0 = Acc[pk], r1 = Acc[pk + 1], x0 = Xc[pk], x1 = Xc[pk + 1];
    const float r0o = __shfl_xor(r0, 32), r1o = __shfl_xor(r1, 32);
    const float x0o = __shfl_xor(x0, 32), x1o = __shfl_xor(x1, 32);
    const float rk = h == hk ? r0 : r0o, rk1 = h == hk ? r1 : r1o; // rows k, k1 of A, by column j
    const float xk = h == hk ? x0 : x0o, xk1 = h == hk ? x1 : x1o; // rows k, k1 of X
    const float akk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rk), k));
    const float ak1k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rk1), k));
    const float ak1k1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rk1), k1));
    fail |= (akk <= 0.f);
    const float i11 = __builtin_amdgcn_rsqf(akk); // 1 / L(k,k)
    const float l21 = ak1k * i11;
    const float d2 = __builtin_fmaf(-l21, l21, ak1k1);
    fail |= (d2 <= 0.f);
    const float i22 = __builtin_amdgcn_rsqf(d2);
    float lk = rk * i11;                             // L(j, k)
    float lk1 = __builtin_fmaf(-l21, lk, rk1) * i22; // L(j, k1)
    lk = j >= k ? lk : 0.f;
    lk1 = j >= k1 ? lk1 : 0.f;
    const float yk = xk * i11;                             // row k of L^-1
    const float yk1 = __builtin_fmaf(-l21, yk, xk1) * i22; // row k1
    if (h == hk) {
      Xc[pk] = yk;
      Xc[pk + 1] = yk1;
    }
    const float xa = h == 0 ? lk : lk1;
    const float xm = j > k1 ? xa : 0.f; // rows <= k1 of X are final
    const float yb = h == 0 ? yk : yk1;
    Acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-xa, xa, Acc, 0, 0, 0);
    Xc = __builtin_amdgcn_mfma_f32_32x32x2f32(-xm, yb, Xc, 0, 0, 0);
  }
  return fail;
}

template <int M>
struct Layout {
  static constexpr int NODE = N * N + N;               // Q | delta
  static constexpr int EDGE = N * N + 2 * N * M + M * M; // A | B | M | R
  static constexpr int VNODE = 2 * N, VEDGE = M, GAIN = M * N + M;
  static constexpr int WSN = N * N + N; // W (C/D-layout dump) | g
};

template <int M>
__global__ __launch_bounds__(64) void chain_factor_solve_mf32(
    const float *__restrict__ mats, const float *__restrict__ vecs, float *__restrict__ sol,
    float *__restrict__ gains, float *__restrict__ wsp, int *__restrict__ status, const long batch,
    const int T
#ifdef SIP_LQR_STAMPS
    , unsigned long long *__restrict__ stamps
#endif
) {
  static_assert(M >= 1 && M <= 8, "control rows must fit regs 0..3 of both halves");
#ifdef SIP_LQR_STAMPS
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
  unsigned long long ts1 = 0;
#endif
  using L = Layout<M>;
  constexpr int STG = L::NODE + L::EDGE, VSTG = L::VNODE + L::VEDGE;
  const long p = blockIdx.x;
  if (p >= batch)
    return;
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const float *pm = mats + p * ((long)(T + 1) * L::NODE + (long)T * L::EDGE);
  const float *pv = vecs + p * ((long)(T + 1) * L::VNODE + (long)T * L::VEDGE);
  float *ps = sol + p * ((long)(T + 1) * L::VNODE + (long)T * L::VEDGE);
  float *pg = gains + p * ((long)T * L::GAIN);
  float *pw = wsp + p * ((long)(T + 1) * L::WSN);

  __shared__ float s_v[N], s_t[N], s_g[N], s_sdi[N], s_h[8], s_k[8], s_x[N], s_z[N], s_u[8];

  int stat = 0;
  f32x16 W, V;

  // ---- node tail: F = I + D^1/2 V D^1/2, Cholesky, W (lqr.cpp:487-529, 722-727)
  // plus the vector terms t = c - delta o v and v for the parent step.
  auto finish_node = [&](const int i) {
    const float *nm = pm + (long)i * STG;
    const float *nv = pv + (long)i * VSTG;
    const float dl = nm[N * N + j];
    if (stat == 0 && __any(dl <= 0.f))
      stat = 1; // INVALID_DELTA
    const float sd = sqrtf(dl), sdi = 1.f / sd;
    if (h == 0) {
      s_sdi[j] = sdi;
      s_x[j] = sd; // scratch: sqrt_delta by row
      const float vj = s_v[j];
      s_t[j] = nv[N + j] - dl * vj; // c - delta o v   (lqr.cpp:778-779, negated)
    }
    __syncthreads();
    // F = I + sd V sd in C/D layout (lqr.cpp:497-503)
    f32x16 Acc, Xc;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool diag = crow(q, h) == j;
      Acc[q] = V[q] * s_x[crow(q, h)] * sd + (diag ? 1.f : 0.f);
      Xc[q] = diag ? 1.f : 0.f; // becomes X = L^-1
    }
    const bool fail = eliminate<16>(Acc, Xc, j, h);
    if (stat == 0 && fail)
      stat = 2; // F_FACTORIZATION_FAILURE
    // F^-1 = X^T X on the matrix pipe, then W = D^-1/2 (I - F^-1) D^-1/2
    f32x16 Finv;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Finv[q] = 0.f;
    prodT<32>(Xc, Xc, Finv);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = crow(q, h);
      W[q] = ((r == j ? 1.f : 0.f) - Finv[q]) * (s_sdi[r] * sdi);
    }
    // spill W (C/D-layout dump) for the rollout
    f32x4 *wd = (f32x4 *)(pw + (long)i * L::WSN);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      wd[g * 64 + lane] = f32x4{W[4 * g], W[4 * g + 1], W[4 * g + 2], W[4 * g + 3]};
    __syncthreads();
  };

  // ---- terminal node -------------------------------------------------------
  V = load_c32(pm + (long)T * STG, N, j, h);
  if (h == 0)
    s_v[j] = pv[(long)T * VSTG + j]; // v = q
  __syncthreads();
  finish_node(T);

  // ---- backward recursion --------------------------------------------------
  for (int i = T - 1; i >= 0; --i) {
    const float *nm = pm + (long)i * STG;
    const float *em = nm + L::NODE;
    const float *nv = pv + (long)i * VSTG;
    const f32x16 A = load_c32(em, N, j, h);
    const f32x16 B = load_c32(em + N * N, M, j, h); // columns >= M are zero
    // g = v_c + W t   (lqr.cpp:778-781)
    {
      const float wt = matTvec(W, s_t, h);
      if (h == 0) {
        const float g = s_v[j] + wt;
        s_g[j] = g;
        pw[(long)(i + 1) * L::WSN + N * N + j] = g;
      }
    }
    // Only rows < M (regs 0..3 of both halves) of the G and H tiles are
    // nonzero: keep just those four registers of each, so that at most five
    // full tiles are ever live (occupancy).
    f32x16 F;
    float G4[4], H4[4];
    const float *Mm = em + N * N + N * M, *Rm = Mm + N * M;
    {
      f32x16 Z, T1;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int r = q + 4 * h; // row of a (<= 8)-row tile for q < 4
        Z[q] = 0.f;
        T1[q] = (q < 4 && r < M && j < M) ? Rm[j * M + r] : 0.f;
      }
      prodT<32>(W, B, Z);  // Z = W B   (H_child^T)
      prodT<32>(B, Z, T1); // G = R + B^T Z
#pragma unroll
      for (int q = 0; q < 4; ++q)
        G4[q] = T1[q];
    }
    {
      f32x16 T2;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int r = q + 4 * h;
        F[q] = 0.f;
        T2[q] = (q < 4 && r < M) ? Mm[r * N + j] : 0.f; // M^T(r, j) = M(j, r)
      }
      prodT<32>(W, A, F);  // F = W A
      prodT<32>(B, F, T2); // H = M^T + B^T F
#pragma unroll
      for (int q = 0; q < 4; ++q)
        H4[q] = T2[q];
    }
    __syncthreads();
    // h = r + B^T g  (lqr.cpp:783-784): value for control row j on lanes j < M
    {
      const float bg = matTvec(B, s_g, h);
      if (lane < M)
        s_h[lane] = nv[L::VNODE + lane] + bg;
    }
    // LLT of G (lqr.cpp:696-701) and G^-1 = X^T X on the m x m corner of a tile
    float K4[4], Gi4[4];
    {
      f32x16 Ga, Gx, Gi;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        Ga[q] = q < 4 ? G4[q] : 0.f;
        Gx[q] = (q < 4 && q + 4 * h == j) ? 1.f : 0.f;
        Gi[q] = 0.f;
      }
      const bool gfail = eliminate<(M + 1) / 2>(Ga, Gx, j, h);
      if (stat == 0 && gfail)
        stat = 3; // G_FACTORIZATION_FAILURE
      if constexpr (M % 2 == 1) { // the padded last column: pivot 0 -> row M of X is garbage
#pragma unroll
        for (int q = 0; q < 4; ++q)
          Gx[q] = (q + 4 * h < M && j < M) ? Gx[q] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        Gi = __builtin_amdgcn_mfma_f32_32x32x2f32(Gx[q], Gx[q], Gi, 0, 0, 0); // G^-1 = X^T X
      // [K | k] = -G^-1 [H | h]   (lqr.cpp:707-713, 785-791); G^-1 symmetric
      f32x16 Kt;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        Kt[q] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Gi4[q] = Gi[q];
        Kt = __builtin_amdgcn_mfma_f32_32x32x2f32(Gi[q], H4[q], Kt, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        K4[q] = -Kt[q]; // K(q + 4h, j)
    }
    __syncthreads();
    float kvj; // k_j on lanes j < M
    {
      float s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        s2 = __builtin_fmaf(Gi4[q], (q + 4 * h < M) ? s_h[q + 4 * h] : 0.f, s2);
      kvj = -(s2 + __shfl_xor(s2, 32));
    }
    // gains out: K (m x 32 col-major) | k
    {
      float *gi = pg + (long)i * L::GAIN;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q + 4 * h < M)
          gi[j * M + q + 4 * h] = K4[q];
      if (lane < M)
        gi[M * N + lane] = kvj;
    }
    // v = q + A^T g + K^T h   (lqr.cpp:793-794)
    {
      float kh = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        kh = __builtin_fmaf(K4[q], (q + 4 * h < M) ? s_h[q + 4 * h] : 0.f, kh);
      kh += __shfl_xor(kh, 32);
      const float vn = nv[j] + matTvec(A, s_g, h) + kh;
      __syncthreads(); // everyone is done with s_v / s_t / s_g of the child
      if (h == 0)
        s_v[j] = vn;
    }
    // V = Q + A^T F + K^T H   (lqr.cpp:715-719)
    V = load_c32(nm, N, j, h);
    prodT<32>(A, F, V);
#pragma unroll
    for (int q = 0; q < 4; ++q) // K^T H: the row pairs (q, q + 4) are the K = 2 slices
      V = __builtin_amdgcn_mfma_f32_32x32x2f32(K4[q], H4[q], V, 0, 0, 0);
    __syncthreads();
    finish_node(i);
  }

#ifdef SIP_LQR_STAMPS
  ts1 = __builtin_amdgcn_s_memtime();
#endif
  // ---- root: g_0 = v_0 + W_0 t_0 ; x_0 = c_0 - delta_0 o g_0, y_0 = g_0 -------
  {
    const float wt = matTvec(W, s_t, h);
    if (h == 0) {
      const float g = s_v[j] + wt;
      const float xx = pv[N + j] - pm[N * N + j] * g;
      ps[j] = xx;
      ps[N + j] = g;
      s_x[j] = xx;
    }
  }
  if (lane == 0)
    status[p] = stat;
  __syncthreads();

  // ---- forward rollout (lqr.cpp:821-870) -------------------------------------
  // Everything stage i + 1 reads from memory is requested while stage i runs
  // (registers): none of it depends on x.
  struct FwdStage {
    float KT[16], AT[16], Bj[8]; // K(j, r(q,h)), A(j, r(q,h)), B(j, 0..M-1) of this lane's row j
    f32x16 Wc;                   // W of the child (C/D-layout dump)
    float kj, gj, cj, dj;        // k_j, g_c(j), c_c(j), delta_c(j)
  };
  auto fetch_stage = [&](const int i, FwdStage &f) {
    const float *em = pm + (long)i * STG + L::NODE;
    const float *nm1 = pm + (long)(i + 1) * STG;
    const float *nv1 = pv + (long)(i + 1) * VSTG;
    const float *gi = pg + (long)i * L::GAIN;
    const float *wn = pw + (long)(i + 1) * L::WSN;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      f.KT[q] = j < M ? gi[crow(q, h) * M + j] : 0.f; // K(j, r) at r * M + j
      f.AT[q] = em[crow(q, h) * N + j];               // A(j, r) at r * 32 + j
    }
#pragma unroll
    for (int r = 0; r < M; ++r)
      f.Bj[r] = em[N * N + r * N + j];
    const f32x4 *wd = (const f32x4 *)wn;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v4 = wd[g * 64 + lane];
      f.Wc[4 * g] = v4[0], f.Wc[4 * g + 1] = v4[1], f.Wc[4 * g + 2] = v4[2], f.Wc[4 * g + 3] = v4[3];
    }
    f.kj = lane < M ? gi[M * N + lane] : 0.f;
    f.gj = wn[N * N + j];
    f.cj = nv1[N + j];
    f.dj = nm1[N * N + j];
  };
  FwdStage cur, nxt;
  if (T > 0)
    fetch_stage(0, cur);
  for (int i = 0; i < T; ++i) {
    if (i + 1 < T)
      fetch_stage(i + 1, nxt);
    // u = k + K x : lane j < M owns control row j
    {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        s = __builtin_fmaf(cur.KT[q], s_x[crow(q, h)], s);
      s += __shfl_xor(s, 32);
      if (lane < M)
        s_u[lane] = cur.kj + s;
    }
    __syncthreads();
    // z = A x + B u : lane j owns state row j
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      z = __builtin_fmaf(cur.AT[q], s_x[crow(q, h)], z);
    z += __shfl_xor(z, 32);
#pragma unroll
    for (int r = 0; r < M; ++r)
      z = __builtin_fmaf(cur.Bj[r], s_u[r], z);
    if (h == 0)
      s_z[j] = z;
    __syncthreads();
    // y_c = g_c + W_c z ; x_c = z + c_c - delta_c o y_c
    const float y = cur.gj + matTvec(cur.Wc, s_z, h);
    const float xn = z + (cur.cj - cur.dj * y);
    float *si = ps + (long)i * VSTG;
    if (lane < M)
      si[2 * N + lane] = s_u[lane];
    if (h == 0) {
      si[VSTG + j] = xn;
      si[VSTG + N + j] = y;
      s_x[j] = xn;
    }
    __syncthreads();
    cur = nxt;
  }
#ifdef SIP_LQR_STAMPS
  if (stamps != nullptr && lane == 0) {
    unsigned long long *o = stamps + (long)blockIdx.x * 24;
    o[0] = ts0, o[1] = ts0, o[2] = ts1, o[3] = ts1, o[4] = __builtin_amdgcn_s_memtime();
  }
#endif
}

} // namespace mf32
} // namespace sipamd
