// kkt_theta_kernels.hpp -- global variables theta (Dimensions::theta_dim = p):
// the Schur-complement part of CallbackProvider::factor / solve
// (helpers.cpp:190-240, 372-407, 896-951) and the theta terms of the KKT
// operator (helpers.cpp:1023-1066, 1128-1158, 1221-1249, 1285-1308, 1344-1367),
// batched.  The stagewise solves in the middle (K^-1 J_theta, K^-1 b) are the
// plan's ordinary solve path, one launch per column of J_theta -- the
// reference's multi-right-hand-side block (helpers.cpp:414-747) computes the
// same quantities column by column with GEMM in place of GEMV.
//
// J_theta and K^-1 J_theta are stored column-major over the batch:
// [col][problem][stagewise_kkt_dim], so each column is a batch of right-hand
// sides / solutions for the stagewise solve.  Completeness path, not tuned.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kkt_kernels.hpp"

namespace sipamd {
namespace kkt {

enum ThetaBlock {
  TH_N_X = 0, TH_N_C, TH_N_G, TH_N_TT, TH_E_X, TH_E_U, TH_E_DYN, TH_E_C, TH_E_G, TH_E_TT, TH_NUM_BLOCKS
};

struct ThetaMeta {
  int p;
  long theta_len;
  const long *to[TH_NUM_BLOCKS];
};

// rows x p block `src` -> rows [at, at + rows) of every column of J (+= if accumulate)
__device__ __forceinline__ void put_rows(double *Jp, long col_stride, int at, const double *src, int rows, int p,
                                         bool accumulate, int tid) {
  for (int idx = tid; idx < rows * p; idx += TPB) {
    const int col = idx / rows, r = idx - col * rows;
    double *dst = Jp + col * col_stride + at + r;
    *dst = (accumulate ? *dst : 0.0) + src[idx];
  }
}

// form_theta_jacobian, helpers.cpp:190-240.  One workgroup per (problem, node): the node's own
// rows and the rows of its child edges (control, child dynamics, edge constraints).
__global__ void __launch_bounds__(TPB)
theta_jacobian_kernel(const Meta mt, const ThetaMeta th, const double *__restrict__ theta_all,
                      double *__restrict__ J_all, long batch) {
  const long prob = blockIdx.x / mt.N;
  const int i = blockIdx.x - (unsigned)(prob * mt.N);
  if (prob >= batch)
    return;
  const int tid = threadIdx.x, p = th.p, sx = mt.x_dim, yd = mt.y_dim;
  const long skkt = (long)sx + yd + mt.z_dim, col_stride = batch * skkt;
  const double *tm = theta_all + prob * th.theta_len;
  double *Jp = J_all + prob * skkt;
  const int n = mt.sd[i];
  put_rows(Jp, col_stride, mt.x_state[i], tm + th.to[TH_N_X][i], n, p, false, tid);
  put_rows(Jp, col_stride, sx + mt.y_node_c[i], tm + th.to[TH_N_C][i], mt.ncd[i], p, false, tid);
  put_rows(Jp, col_stride, sx + yd + mt.z_node[i], tm + th.to[TH_N_G][i], mt.ngd[i], p, false, tid);
  if (mt.in_edge[i] < 0) { // the root's dynamics rows stay zero (J_theta.setZero(), :198)
    for (int idx = tid; idx < n * p; idx += TPB) {
      const int col = idx / n, r = idx - col * n;
      Jp[col * col_stride + sx + mt.y_dyn[i] + r] = 0.0;
    }
  }
  for (int ci = mt.child_offsets[i]; ci < mt.child_offsets[i + 1]; ++ci) {
    const int e = mt.child_edges[ci], ch = mt.child[e];
    __syncthreads(); // the accumulation below reads what this workgroup wrote above
    put_rows(Jp, col_stride, mt.x_state[i], tm + th.to[TH_E_X][e], n, p, true, tid); // :222-223
    put_rows(Jp, col_stride, mt.x_control[e], tm + th.to[TH_E_U][e], mt.cd[e], p, false, tid);
    put_rows(Jp, col_stride, sx + mt.y_dyn[ch], tm + th.to[TH_E_DYN][e], mt.sd[ch], p, false, tid);
    put_rows(Jp, col_stride, sx + mt.y_edge_c[e], tm + th.to[TH_E_C][e], mt.ecd[e], p, false, tid);
    put_rows(Jp, col_stride, sx + yd + mt.z_edge[e], tm + th.to[TH_E_G][e], mt.egd[e], p, false, tid);
  }
}

// S = sum d2L_dtheta2 + diag(r1_theta) - J^T (K^-1 J), then LLT in place (helpers.cpp:389-407).
// One workgroup per problem; lane q owns entry (row a, col b) = (q % p, q / p).
__global__ void __launch_bounds__(TPB)
theta_schur_kernel(const Meta mt, const ThetaMeta th, const double *__restrict__ theta_all,
                   const double *__restrict__ r1_all, const double *__restrict__ J_all,
                   const double *__restrict__ KJ_all, double *__restrict__ S_all, int32_t *__restrict__ status,
                   long batch, int fail_code) {
  extern __shared__ double sm[]; // p * p
  const long prob = blockIdx.x;
  if (prob >= batch || status[prob] != 0)
    return;
  const int tid = threadIdx.x, p = th.p, sx = mt.x_dim;
  const long skkt = (long)sx + mt.y_dim + mt.z_dim, col_stride = batch * skkt;
  const double *tm = theta_all + prob * th.theta_len;
  const double *Jp = J_all + prob * skkt, *KJp = KJ_all + prob * skkt;
  for (int q = tid; q < p * p; q += TPB) {
    const int b = q / p, a = q - b * p;
    double acc = 0.0;
    for (int i = 0; i < mt.N; ++i)
      acc += (tm + th.to[TH_N_TT][i])[q];
    for (int e = 0; e < mt.E; ++e)
      acc += (tm + th.to[TH_E_TT][e])[q];
    if (a == b)
      acc += r1_all[prob * (sx + p) + sx + a];
    const double *ja = Jp + a * col_stride, *kb = KJp + b * col_stride;
    double dot = 0.0;
    for (long r = 0; r < skkt; ++r)
      dot += ja[r] * kb[r];
    sm[q] = acc - dot;
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
  for (int q = tid; q < p * p; q += TPB)
    S_all[prob * p * p + q] = sm[q];
}

// Same, for p <= PMAX: the lanes stride over the rows r of J / K^-1 J (coalesced: consecutive rows of
// a column are contiguous) and keep all p x p partial sums in registers; one butterfly reduction at
// the end.  (The generic kernel above walks each (a, b) pair's two columns with one lane.)
template <int PMAX>
__global__ void __launch_bounds__(TPB)
theta_schur_small_kernel(const Meta mt, const ThetaMeta th, const double *__restrict__ theta_all,
                         const double *__restrict__ r1_all, const double *__restrict__ J_all,
                         const double *__restrict__ KJ_all, double *__restrict__ S_all,
                         int32_t *__restrict__ status, long batch, int fail_code) {
  __shared__ double sm[PMAX * PMAX];
  const long prob = blockIdx.x;
  if (prob >= batch || status[prob] != 0)
    return;
  const int tid = threadIdx.x, p = th.p, sx = mt.x_dim;
  const long skkt = (long)sx + mt.y_dim + mt.z_dim, col_stride = batch * skkt;
  const double *tm = theta_all + prob * th.theta_len;
  const double *Jp = J_all + prob * skkt, *KJp = KJ_all + prob * skkt;
  double acc[PMAX][PMAX];
#pragma unroll
  for (int a = 0; a < PMAX; ++a)
#pragma unroll
    for (int b = 0; b < PMAX; ++b)
      acc[a][b] = 0.0;
  for (long r = tid; r < skkt; r += TPB) {
    double ja[PMAX], kb[PMAX];
#pragma unroll
    for (int a = 0; a < PMAX; ++a) {
      ja[a] = a < p ? Jp[a * col_stride + r] : 0.0;
      kb[a] = a < p ? KJp[a * col_stride + r] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < PMAX; ++a)
#pragma unroll
      for (int b = 0; b < PMAX; ++b)
        acc[a][b] += ja[a] * kb[b];
  }
#pragma unroll
  for (int a = 0; a < PMAX; ++a)
#pragma unroll
    for (int b = 0; b < PMAX; ++b) {
      double v = acc[a][b];
      for (int off = TPB / 2; off > 0; off >>= 1)
        v += __shfl_xor(v, off, TPB);
      if (tid == 0 && a < p && b < p)
        sm[a + p * b] = v;
    }
  __syncthreads();
  for (int q = tid; q < p * p; q += TPB) {
    const int b = q / p, a = q - b * p;
    double base = 0.0;
    for (int i = 0; i < mt.N; ++i)
      base += (tm + th.to[TH_N_TT][i])[q];
    for (int e = 0; e < mt.E; ++e)
      base += (tm + th.to[TH_E_TT][e])[q];
    if (a == b)
      base += r1_all[prob * (sx + p) + sx + a];
    sm[q] = base - sm[q];
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
  for (int q = tid; q < p * p; q += TPB)
    S_all[prob * p * p + q] = sm[q];
}

// b = [x | theta | y | z] -> stagewise [x | y | z]  (helpers.cpp:911-916)
__global__ void __launch_bounds__(256)
theta_strip_kernel(const double *__restrict__ b_all, double *__restrict__ out_all, int sx, int p, long skkt,
                   long batch) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= batch * skkt)
    return;
  const long prob = idx / skkt;
  const long r = idx - prob * skkt;
  out_all[idx] = b_all[prob * (skkt + p) + (r < sx ? r : r + p)];
}

// theta = S^-1 (b_theta - J^T K^-1 b); sol = K^-1 b - (K^-1 J) theta, re-inserted as
// [x | theta | y | z]  (helpers.cpp:920-950).  One workgroup per problem.
__global__ void __launch_bounds__(TPB)
theta_finish_kernel(const Meta mt, const ThetaMeta th, const double *__restrict__ b_all,
                    const double *__restrict__ J_all, const double *__restrict__ KJ_all,
                    const double *__restrict__ S_all, const double *__restrict__ sw_all,
                    double *__restrict__ sol_all, const int32_t *__restrict__ status, long batch) {
  extern __shared__ double sm[]; // theta (p)
  const long prob = blockIdx.x;
  if (prob >= batch || status[prob] != 0)
    return;
  const int tid = threadIdx.x, p = th.p, sx = mt.x_dim;
  const long skkt = (long)sx + mt.y_dim + mt.z_dim, col_stride = batch * skkt;
  const double *Jp = J_all + prob * skkt, *KJp = KJ_all + prob * skkt, *sw = sw_all + prob * skkt;
  const double *b_theta = b_all + prob * (skkt + p) + sx;
  double *sol = sol_all + prob * (skkt + p);
  for (int a = 0; a < p; ++a) { // J_theta^T K^-1 b: lanes over the rows (coalesced), butterfly sum
    const double *ja = Jp + a * col_stride;
    double dot = 0.0;
    for (long r = tid; r < skkt; r += TPB)
      dot += ja[r] * sw[r];
    for (int off = TPB / 2; off > 0; off >>= 1)
      dot += __shfl_xor(dot, off, TPB);
    if (tid == 0)
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

// Theta terms of y += K x (or of the selected blocks of it, ApplyIO::parts: the theta sections of
// add_Hx / Cx / CTx / Gx / GTx_to_y, helpers.cpp:1023-1066, 1128-1158, 1221-1249, 1285-1308,
// 1344-1367).  x-space vectors are [stagewise x | theta].  One workgroup per problem; lane a < p
// accumulates y_theta[a]; the rows coupled to theta are updated by the lanes in turn.
__global__ void __launch_bounds__(TPB)
apply_theta_kernel(const Meta mt, const ThetaMeta th, const double *__restrict__ theta_all,
                   const double *__restrict__ r1_all, const ApplyIO io, long batch) {
  const long prob = blockIdx.x;
  if (prob >= batch)
    return;
  const int tid = threadIdx.x, p = th.p, sx = mt.x_dim;
  const int parts = io.parts;
  const bool pH = parts & AP_H, pC = parts & AP_C, pCT = parts & AP_CT, pG = parts & AP_G, pGT = parts & AP_GT,
             pR = parts & AP_REG;
  const double *tm = theta_all + prob * th.theta_len;
  const double *x_x = io.x_x ? io.x_x + prob * io.sx : nullptr, *x_y = io.x_y ? io.x_y + prob * io.sy : nullptr;
  const double *x_z = io.x_z ? io.x_z + prob * io.sz : nullptr;
  double *y_x = io.y_x ? io.y_x + prob * io.sx : nullptr, *y_y = io.y_y ? io.y_y + prob * io.sy : nullptr;
  double *y_z = io.y_z ? io.y_z + prob * io.sz : nullptr;
  const double *theta = x_x ? x_x + sx : nullptr; // read by H, C, G (and the r1 term)
  double *y_theta = y_x ? y_x + sx : nullptr;     // written by H, CT, GT (and the r1 term)
  // rows += M theta (M rows x p); lanes over rows
  auto add_M_theta = [&](double *dst, const double *M, int rows) {
    for (int r = tid; r < rows; r += TPB) {
      double acc = 0.0;
      for (int a = 0; a < p; ++a)
        acc += M[r + (long)rows * a] * theta[a];
      dst[r] += acc;
    }
  };
  // y_theta[a] += sum_r M[r, a] v[r]; lane a
  double yt = 0.0;
  auto add_MT_v = [&](const double *M, int rows, const double *v) {
    if (tid < p) {
      double acc = 0.0;
      for (int r = 0; r < rows; ++r)
        acc += M[r + (long)rows * tid] * v[r];
      yt += acc;
    }
  };
  auto add_M_own = [&](const double *H) { // y_theta[a] += sum_b H[a, b] theta[b]
    if (tid < p) {
      double acc = 0.0;
      for (int b = 0; b < p; ++b)
        acc += H[tid + (long)p * b] * theta[b];
      yt += acc;
    }
  };
  for (int i = 0; i < mt.N; ++i) {
    const int n = mt.sd[i], c = mt.ncd[i], g = mt.ngd[i];
    const double *Hxt = tm + th.to[TH_N_X][i];
    if (pH) {
      add_M_theta(y_x + mt.x_state[i], Hxt, n);
      add_MT_v(Hxt, n, x_x + mt.x_state[i]);
      add_M_own(tm + th.to[TH_N_TT][i]); // y_theta += H_theta_theta theta (:1038-1040)
    }
    if (pC)
      add_M_theta(y_y + mt.y_node_c[i], tm + th.to[TH_N_C][i], c);
    if (pCT)
      add_MT_v(tm + th.to[TH_N_C][i], c, x_y + mt.y_node_c[i]);
    if (pG)
      add_M_theta(y_z + mt.z_node[i], tm + th.to[TH_N_G][i], g);
    if (pGT)
      add_MT_v(tm + th.to[TH_N_G][i], g, x_z + mt.z_node[i]);
    __syncthreads(); // edge terms below add to the same state rows
  }
  for (int e = 0; e < mt.E; ++e) {
    const int pa = mt.parent[e], ch = mt.child[e];
    const int n = mt.sd[pa], nc = mt.sd[ch], m = mt.cd[e], c = mt.ecd[e], g = mt.egd[e];
    if (pH) {
      add_M_theta(y_x + mt.x_state[pa], tm + th.to[TH_E_X][e], n);
      add_M_theta(y_x + mt.x_control[e], tm + th.to[TH_E_U][e], m);
      add_MT_v(tm + th.to[TH_E_X][e], n, x_x + mt.x_state[pa]);
      add_MT_v(tm + th.to[TH_E_U][e], m, x_x + mt.x_control[e]);
      add_M_own(tm + th.to[TH_E_TT][e]);
    }
    if (pC)
      add_M_theta(y_y + mt.y_dyn[ch], tm + th.to[TH_E_DYN][e], nc);
    if (pCT)
      add_MT_v(tm + th.to[TH_E_DYN][e], nc, x_y + mt.y_dyn[ch]);
    if (pC)
      add_M_theta(y_y + mt.y_edge_c[e], tm + th.to[TH_E_C][e], c);
    if (pCT)
      add_MT_v(tm + th.to[TH_E_C][e], c, x_y + mt.y_edge_c[e]);
    if (pG)
      add_M_theta(y_z + mt.z_edge[e], tm + th.to[TH_E_G][e], g);
    if (pGT)
      add_MT_v(tm + th.to[TH_E_G][e], g, x_z + mt.z_edge[e]);
    __syncthreads(); // sibling edges add to the same parent rows
  }
  if (tid < p && (pH || pCT || pGT || pR))
    y_theta[tid] += pR ? yt + r1_all[prob * (sx + p) + sx + tid] * theta[tid] : yt;
}

} // namespace kkt
} // namespace sipamd
