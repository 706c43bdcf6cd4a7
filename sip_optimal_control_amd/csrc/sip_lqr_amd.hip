// sip_lqr_amd.hip -- C ABI (include/sip_lqr_amd.h) over the gfx950 kernels.
// Host side of the drop-in boundary; no CPU compute path exists here.
#include "../../include/sip_lqr_amd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <vector>

#include <memory>

#include "chain_mf32.hpp"
#include "mt16_launch.hpp"
#include "generic_plan.hpp"
#include "qw16_launch.hpp"
#include "stream_fill.hpp"

struct sip_lqr_plan {
  int layout = SIP_LQR_LAYOUT_FULL; // of mats
  int dtype;
  int64_t batch;
  int T, n, m, device;
  const char *kernel_name;
  int ws_slot; // scalars of workspace per node of the fused kernel (0: none)
  // fused factor+solve launcher of a dedicated kernel; nullptr: the general
  // engine (tree_generic.hpp) runs factor then solve
  sipamd::launch_fs_t launch_fs;
  // several right-hand sides per sweep (chain_mrhs.hpp); nullptr: column by column
  sipamd::launch_mrhs_t launch_mrhs = nullptr;
  // General engine on the packed chain layout: serves shapes / dtypes without
  // a dedicated kernel and the split factor / solve entry points.  Tables are
  // laid out and uploaded at plan creation, so that the compute entry points
  // only enqueue kernels (graph-capturable from the first call on).
  sipamd::GenericPlan gen;
  // false: the plan was created on a host without a HIP device and serves the
  // host-side entry points only (sizes, pack / unpack, kernel name).
  bool on_device = false;
  // Split factor / solve of a shape with a fused kernel: both re-run the fused sweep (factor on a
  // zero right-hand side), which beats a second, slower kernel family by an order of magnitude;
  // SIP_LQR_SPLIT=general keeps them on the general engine.
  bool split_on_fused = false;
  // Uniform shapes without an exact kernel run on the next larger fused kernel (kn >= n, km >= m):
  // the problem is embedded in a kn x km one whose extra states / controls decouple exactly
  // (Q = I, R = I, delta = 1 on their diagonal, zeros elsewhere: x = y = u = 0 there), by a
  // device-side repack before and after the sweep.  SIP_LQR_PAD=0 keeps the general engine.
  bool padded = false;
  int kn = 0, km = 0;
  bool solve_only = false; // the kernel has the vector-only solve mode (qw16)
  std::string name_storage;
};

// Overridden by the object __graft_entry__.build_hip() generates (build_stamp.c); a library linked
// without it (tools/diag_build.sh) says so.
extern "C" __attribute__((weak)) const char sip_lqr_build_stamp[] = "SIPLQRSRC=unstamped";

#ifdef SIP_LQR_STAMPS
// Diagnostic build: device buffer of 8 x u64 per wave, set by the tool.
unsigned long long *g_sip_lqr_stamps = nullptr;
extern "C" void sip_lqr_debug_set_stamps(void *p) { g_sip_lqr_stamps = (unsigned long long *)p; }
#endif

namespace {

template <int M>
hipError_t launch_mf32(long batch, int T, const void *mats, const void *vecs, void *sol, void *gains,
                       int32_t *status, void *ws, hipStream_t stream, int /*mode: always the full sweep*/,
                       void * /*gfac*/) {
  hipLaunchKernelGGL((sipamd::mf32::chain_factor_solve_mf32<M>), dim3((unsigned)batch), dim3(64), 0,
                     stream, (const float *)mats, (const float *)vecs, (float *)sol, (float *)gains,
                     (float *)ws, (int *)status, batch, T SIP_STAMP_PASS);
  return hipGetLastError();
}

using sipamd::KernelEntry;

// First match wins; SIP_LQR_VARIANT=direct|staged (tests, A/B timing) narrows
// the search to kernels whose name carries that tag.
// fp32, n = 32: one problem per wavefront, products on the matrix cores
#define MF32(M)                                                                \
  { SIP_LQR_F32, 32, M, "chain_factor_solve_mf32<32," #M ",mfma>/f32",          \
    sipamd::mf32::Layout<M>::WSN, &launch_mf32<M> }

const KernelEntry kKernels[] = {
#if defined(SIP_QW16_QUICK) && defined(SIP_QW16_QUICK_C4) // tools/ab_build.sh: one kernel alone, for A/B timing
    MT16_ENTRY(SIP_LQR_F32, float, "f32", 8), MF32(8),
#elif defined(SIP_QW16_QUICK) && defined(SIP_QW16_QUICK_DIRECT)
    QW16_STAGED(14, 8), QW16_STAGED(15, 4), QW16_STAGED(13, 5), QW16_STAGED(12, 8), QW16_STAGED(14, 4), QW16_STAGED(12, 6),
#elif defined(SIP_QW16_QUICK) && defined(SIP_QW16_QUICK_MR) // ... with its multi-right-hand-side solve
    QW16_STAGED_MR(12, 4),
#elif defined(SIP_QW16_QUICK)
    QW16_STAGED(12, 4),
#else
    // n = 32 on 16 x 16 matrix-core tiles (chain_mt16.hpp), fp32 and fp64; the 32x32x2 kernel of round 1
    // (chain_mf32.hpp) stays selectable by SIP_LQR_VARIANT=mf32 for A/B timing
    MT16_ENTRY(SIP_LQR_F32, float, "f32", 8), MT16_ENTRY(SIP_LQR_F32, float, "f32", 4),
    MT16_ENTRY(SIP_LQR_F64, double, "f64", 8), MT16_ENTRY(SIP_LQR_F64, double, "f64", 4),
    // symmetric-packed layout (sip_lqr_plan_create_layout): found only by plans that ask for it
    QW16_STAGED_SYM(12, 4), QW16_STAGED_SYM(8, 4), QW16_STAGED_SYM(4, 4),
    MF32(8), QW16_STAGED_MR(12, 4), QW16_STAGED_MR(4, 2), QW16_DIRECT_MR(12, 4),
    QW16_DIRECT_MR(4, 2),  QW16_STAGED_MR(1, 1), QW16_STAGED_MR(2, 1),
    QW16_STAGED_MR(3, 2),  QW16_STAGED_MR(8, 3),
    // the grid of the reference's benchmarks (lqr_benchmark.cpp:537-545,
    // newton_kkt_benchmark.cpp:264-273: n in {4, 6, 8}, m in {1, 2, 3, 4}; n = 16 has no
    // vector lane left and runs on the general engine) and n = 12 with fewer controls
    QW16_STAGED_MR(4, 4),  QW16_STAGED_MR(6, 2),  QW16_STAGED_MR(6, 4),  QW16_STAGED_MR(8, 2),
    QW16_STAGED_MR(8, 4),  QW16_STAGED_MR(12, 2), QW16_STAGED_MR(4, 1),  QW16_STAGED_MR(4, 3),
    QW16_STAGED_MR(6, 1),  QW16_STAGED_MR(6, 3),  QW16_STAGED_MR(8, 1),  QW16_STAGED_MR(12, 1),
    QW16_STAGED_MR(12, 3),
    // hosts for the embedding of larger shapes (n <= 15: one lane of the row carries the affine column)
    QW16_STAGED_MR(8, 8),  QW16_STAGED_MR(12, 8), QW16_STAGED_MR(14, 4), QW16_STAGED_MR(14, 8), QW16_STAGED_MR(15, 4),
    QW16_STAGED_MR(15, 8),
    // n = 16 (in the reference's benchmark grid): distributed-vector mode, see chain_qw16.hpp
    QW16_DIRECT_MR(16, 1), QW16_DIRECT_MR(16, 2), QW16_DIRECT_MR(16, 3), QW16_DIRECT_MR(16, 4), QW16_DIRECT_MR(16, 8),
#endif
};

// Every other shape n <= 16, m <= 8 (qw16_extra.hip, compiled in SIP_QW16_SLICES slices).
// SIP_LQR_EXTRA=0 (tests of the embedding) hides them.
template <typename F> void for_each_kernel(F &&f) {
  for (const auto &k : kKernels)
    f(k);
#ifndef SIP_QW16_NO_EXTRA
  const char *extra = std::getenv("SIP_LQR_EXTRA");
  if (extra != nullptr && extra[0] == '0')
    return;
  typedef const KernelEntry *(*slice_fn)(int *);
  static const slice_fn slices[sipamd::kQw16ExtraSlices] = {
      sipamd::qw16_extra_slice_0, sipamd::qw16_extra_slice_1, sipamd::qw16_extra_slice_2, sipamd::qw16_extra_slice_3,
      sipamd::qw16_extra_slice_4, sipamd::qw16_extra_slice_5, sipamd::qw16_extra_slice_6, sipamd::qw16_extra_slice_7};
  for (const slice_fn fn : slices) {
    int count = 0;
    const KernelEntry *table = fn(&count);
    for (int e = 0; e < count; ++e)
      f(table[e]);
  }
#endif
}

const KernelEntry *find_kernel(int dtype, int n, int m, int layout = SIP_LQR_LAYOUT_FULL) {
  const char *want = std::getenv("SIP_LQR_VARIANT");
  const KernelEntry *found = nullptr;
  for_each_kernel([&](const KernelEntry &k) {
    if (found == nullptr && k.dtype == dtype && k.n == n && k.m == m && k.layout == layout &&
        (want == nullptr || want[0] == 0 || std::strstr(k.name, want)))
      found = &k;
  });
  return found;
}

// Smallest fused fp64 kernel that can embed an (n, m) chain: least N, then a staged kernel with
// the least M, then a direct one.
const KernelEntry *find_embedding_kernel(int n, int m) {
  const KernelEntry *best = nullptr;
  auto better = [&](const KernelEntry &k) {
    if (best == nullptr)
      return true;
    if (k.n != best->n)
      return k.n < best->n;
    const bool ks = std::strstr(k.name, "staged") != nullptr, bs = std::strstr(best->name, "staged") != nullptr;
    if (ks != bs)
      return ks;
    return k.m < best->m;
  };
  for (const auto &k : kKernels)
    if (k.dtype == SIP_LQR_F64 && k.layout == SIP_LQR_LAYOUT_FULL && k.n >= n && k.m >= m && better(k))
      best = &k;
  return best;
}

size_t scalar_size(const sip_lqr_plan *p) {
  return p->dtype == SIP_LQR_F32 ? sizeof(float) : sizeof(double);
}

// Chain topology (Topology::set_chain, lqr.cpp:32-40) for the general engine.
void init_generic(sip_lqr_plan *p) {
  const int T = p->T, N = T + 1;
  std::vector<int> sd(N, p->n), cd(T, p->m), pa(T), ch(T), marks(N);
  for (int e = 0; e < T; ++e)
    pa[e] = e, ch[e] = e + 1;
  sipamd::GenericPlan &g = p->gen;
  g.set_shape(T, 0, sd.data(), cd.data());
  g.parents.assign(T, 0), g.children.assign(T, 0);
  g.child_offsets.assign(N + 1, 0), g.child_edges.assign(T, 0);
  g.preorder.assign(N, 0), g.postorder.assign(N, 0);
  (void)sip_lqr_compile_topology(T, 0, pa.data(), ch.data(), g.child_offsets.data(),
                                 g.child_edges.data(), g.parents.data(), g.children.data(),
                                 g.preorder.data(), g.postorder.data(), marks.data());
  g.layout_chain(p->n, p->m, T);
}

// Compute entry points: the plan must live on a device (see sip_lqr_plan::on_device).
hipError_t require_device(const sip_lqr_plan *p) { return p->on_device ? hipSuccess : hipErrorNoDevice; }

// Bytes of the general engine's work arena, plus the status copy it keeps for
// sip_lqr_solve (which has no status argument).
size_t generic_ws_bytes(const sip_lqr_plan *p) {
  const size_t body = (size_t)p->batch * (size_t)p->gen.ws_len * scalar_size(p);
  return (body + 15) / 16 * 16 + (size_t)p->batch * sizeof(int32_t);
}
int32_t *generic_status(const sip_lqr_plan *p, void *ws) {
  const size_t body = (size_t)p->batch * (size_t)p->gen.ws_len * scalar_size(p);
  return (int32_t *)((char *)ws + (body + 15) / 16 * 16);
}

// ---- embedding of an (n, m) chain in a (N, M) one -----------------------------------------
struct PadDims {
  int n, m, N, M, T;
  long mats_len, vecs_len, gains_len, pmats_len, pvecs_len, pgains_len;
};
PadDims pad_dims(const sip_lqr_plan *p) {
  PadDims d;
  d.n = p->n, d.m = p->m, d.N = p->kn, d.M = p->km, d.T = p->T;
  const long n = d.n, m = d.m, N = d.N, M = d.M, T = d.T;
  d.mats_len = (T + 1) * (n * n + n) + T * (n * n + 2 * n * m + m * m);
  d.vecs_len = (T + 1) * 2 * n + T * m;
  d.gains_len = T * (m * n + m);
  d.pmats_len = (T + 1) * (N * N + N) + T * (N * N + 2 * N * M + M * M);
  d.pvecs_len = (T + 1) * 2 * N + T * M;
  d.pgains_len = T * (M * N + M);
  return d;
}

// blockIdx.y: problem (strided), blockIdx.x: stage; the threads sweep the padded stage block.
// Small-integer quotients via float reciprocals (exact for the sizes involved, < 2^12).
__device__ __forceinline__ int div_small(const int v, const float inv) { return (int)(((float)v + 0.5f) * inv); }

__global__ void __launch_bounds__(256)
pad_mats_kernel(const PadDims d, const double *__restrict__ mats, double *__restrict__ pm, long batch) {
  const int n = d.n, m = d.m, N = d.N, M = d.M;
  const int i = blockIdx.x; // stage (the terminal stage has only the node block)
  const int pnode = N * N + N, pstage = pnode + N * N + 2 * N * M + M * M;
  const int stage = n * n + n + n * n + 2 * n * m + m * m;
  const int count = i < d.T ? pstage : pnode;
  const float invN = 1.0f / (float)N, invM = 1.0f / (float)M;
  for (long prob = blockIdx.y; prob < batch; prob += gridDim.y) {
    const double *src = mats + prob * d.mats_len + (long)i * stage;
    double *dst = pm + prob * d.pmats_len + (long)i * pstage;
    for (int at = threadIdx.x; at < count; at += blockDim.x) {
      double v;
      int o = at;
      if (o < N * N) { // Q: identity on the extra diagonal
        const int col = div_small(o, invN), row = o - col * N;
        v = (row < n && col < n) ? src[row + n * col] : (row == col ? 1.0 : 0.0);
      } else if ((o -= N * N) < N) { // delta: 1 on the extra states
        v = o < n ? src[n * n + o] : 1.0;
      } else if ((o -= N) < N * N) { // A
        const int col = div_small(o, invN), row = o - col * N;
        v = (row < n && col < n) ? src[n * n + n + row + n * col] : 0.0;
      } else if ((o -= N * N) < N * M) { // B
        const int col = div_small(o, invN), row = o - col * N;
        v = (row < n && col < m) ? src[2 * n * n + n + row + n * col] : 0.0;
      } else if ((o -= N * M) < N * M) { // M (cross term)
        const int col = div_small(o, invN), row = o - col * N;
        v = (row < n && col < m) ? src[2 * n * n + n + n * m + row + n * col] : 0.0;
      } else { // R: identity on the extra diagonal
        o -= N * M;
        const int col = div_small(o, invM), row = o - col * M;
        v = (row < m && col < m) ? src[2 * n * n + n + 2 * n * m + row + m * col] : (row == col ? 1.0 : 0.0);
      }
      dst[at] = v;
    }
  }
}

// vecs -> padded vecs (zeros on the extras); PAD = false: padded sol -> sol
template <bool PAD>
__global__ void __launch_bounds__(256)
pad_vecs_kernel(const PadDims d, const double *__restrict__ src_all, double *__restrict__ dst_all, long batch) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long len = PAD ? d.pvecs_len : d.vecs_len;
  if (idx >= batch * len)
    return;
  const long prob = idx / len;
  long at = idx - prob * len;
  const int n = d.n, m = d.m, N = d.N, M = d.M;
  const int a = PAD ? N : n, b = PAD ? M : m;   // dims of the layout this thread indexes
  const int oa = PAD ? n : N, ob = PAD ? m : M; // dims of the other layout
  const int i = (int)(at / (2 * a + b));
  const int o = (int)(at - (long)i * (2 * a + b));
  int other = -1; // offset inside the other layout's stage, -1: an extra (padded) entry
  if (o < a)
    other = o < n ? o : -1;
  else if (o < 2 * a)
    other = (o - a) < n ? oa + (o - a) : -1;
  else
    other = (o - 2 * a) < m ? 2 * oa + (o - 2 * a) : -1;
  const long other_at = prob * (PAD ? d.vecs_len : d.pvecs_len) + (long)i * (2 * oa + ob) + other;
  dst_all[idx] = other >= 0 ? src_all[other_at] : 0.0;
}

// padded gains (K: M x N, k: M) -> gains (K: m x n, k: m)
__global__ void __launch_bounds__(256)
unpad_gains_kernel(const PadDims d, const double *__restrict__ pg, double *__restrict__ gains, long batch) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= batch * d.gains_len)
    return;
  const long prob = idx / d.gains_len;
  long at = idx - prob * d.gains_len;
  const int n = d.n, m = d.m, N = d.N, M = d.M;
  const int i = (int)(at / (m * n + m));
  const int o = (int)(at - (long)i * (m * n + m));
  const double *src = pg + prob * d.pgains_len + (long)i * (M * N + M);
  if (o < m * n) {
    const int col = o / m, row = o - col * m;
    gains[idx] = src[row + M * col];
  } else {
    gains[idx] = src[M * N + (o - m * n)];
  }
}

// Workspace of the fused kernel: spill | scratch vecs (zero rhs of factor) | scratch sol |
// status copy (for sip_lqr_solve, which has no status argument).
struct FusedSplit {
  size_t vecs, sol, status, gfac, pmats, pgains, total;
};
FusedSplit fused_split_layout(const sip_lqr_plan *p) {
  const size_t N = p->kn, M = p->km, T = p->T, B = (size_t)p->batch, sz = scalar_size(p);
  const size_t spill = B * (T + 1) * (size_t)p->ws_slot * sz;
  const size_t vb = B * ((T + 1) * 2 * N + T * M) * sz; // vecs / sol in the kernel's dimensions
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const bool scratch = p->split_on_fused || p->padded;
  FusedSplit f;
  f.vecs = up(spill);
  f.sol = f.vecs + (scratch ? up(vb) : 0);
  f.status = f.sol + (scratch ? up(vb) : 0);
  f.gfac = f.status + (scratch ? up(B * sizeof(int32_t)) : 0); // G factors of the split factor (mode 1)
  f.pmats = f.gfac + (p->split_on_fused ? up(B * T * (M * M + M) * sz) : 0);
  f.pgains = f.pmats + (p->padded ? up(B * ((T + 1) * (N * N + N) + T * (N * N + 2 * N * M + M * M)) * sz) : 0);
  f.total = f.pgains + (p->padded ? up(B * T * (M * N + M) * sz) : 0);
  return f;
}

// The sweep of a plan, through the embedding when the plan is padded.  mode 0: fused factor + solve;
// 1: split factor (vecs == nullptr: zero right-hand side; sol not wanted); 2: split solve against
// the state a mode-1 call left in `ws` (and, unpadded, in `gains`); kernels without a solve-only
// mode run the full sweep instead.
hipError_t run_fused(const sip_lqr_plan *p, const void *mats, const void *vecs, void *sol, void *gains,
                     int32_t *status, void *ws, hipStream_t s, int mode) {
  const FusedSplit f = fused_split_layout(p);
  char *w = (char *)ws;
  hipError_t e = hipSuccess;
  if (mode == 2 && !p->solve_only)
    mode = 0;
  if (!p->padded) {
    const void *v = vecs;
    if (v == nullptr) {
      e = sipamd::zero_async(w + f.vecs, f.sol - f.vecs, s);
      v = w + f.vecs;
    }
    if (e == hipSuccess)
      e = p->launch_fs(p->batch, p->T, mats, v, sol ? sol : (void *)(w + f.sol), gains, status, ws, s, mode, w + f.gfac);
    return e;
  }
  const PadDims d = pad_dims(p);
  const long B = p->batch;
  auto grid = [](long count) { return dim3((unsigned)((count + 255) / 256)); };
  if (mode != 2) // the padded matrices (and gains) of the factor call are still in the workspace
    hipLaunchKernelGGL(pad_mats_kernel, dim3((unsigned)(d.T + 1), (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, s,
                       d, (const double *)mats, (double *)(w + f.pmats), B);
  if (vecs != nullptr)
    hipLaunchKernelGGL(pad_vecs_kernel<true>, grid(B * d.pvecs_len), dim3(256), 0, s, d, (const double *)vecs,
                       (double *)(w + f.vecs), B);
  else
    e = sipamd::zero_async(w + f.vecs, f.sol - f.vecs, s);
  if (e == hipSuccess)
    e = p->launch_fs(p->batch, p->T, w + f.pmats, w + f.vecs, w + f.sol, w + f.pgains, status, ws, s, mode, w + f.gfac);
  if (e == hipSuccess && sol != nullptr)
    hipLaunchKernelGGL(pad_vecs_kernel<false>, grid(B * d.vecs_len), dim3(256), 0, s, d, (const double *)(w + f.sol),
                       (double *)sol, B);
  if (e == hipSuccess && gains != nullptr && d.gains_len > 0)
    hipLaunchKernelGGL(unpad_gains_kernel, grid(B * d.gains_len), dim3(256), 0, s, d,
                       (const double *)(w + f.pgains), (double *)gains, B);
  return e == hipSuccess ? hipGetLastError() : e;
}

int report(hipError_t e, const char *what) {
  if (e == hipSuccess)
    return SIP_LQR_OK;
  std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
  return SIP_LQR_ERR_HIP;
}

} // namespace

extern "C" {

int sip_lqr_plan_create(int dtype, int64_t batch, int T, int n, int m,
                        int device, sip_lqr_plan **plan) {
  return sip_lqr_plan_create_layout(dtype, batch, T, n, m, device, SIP_LQR_LAYOUT_FULL, plan);
}

int sip_lqr_plan_layout(const sip_lqr_plan *plan) { return plan ? plan->layout : SIP_LQR_LAYOUT_FULL; }

int sip_lqr_plan_create_layout(int dtype, int64_t batch, int T, int n, int m, int device, int layout,
                               sip_lqr_plan **plan) {
  if (plan == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  *plan = nullptr;
  if (batch < 1 || T < 0 || n < 1 || m < 1 ||
      (dtype != SIP_LQR_F64 && dtype != SIP_LQR_F32) ||
      (layout != SIP_LQR_LAYOUT_FULL && layout != SIP_LQR_LAYOUT_SYMMETRIC))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const char *want = std::getenv("SIP_LQR_VARIANT");
  const bool force_general = want != nullptr && std::strcmp(want, "general") == 0;
  const KernelEntry *k = force_general ? nullptr : find_kernel(dtype, n, m, layout);
  if (layout != SIP_LQR_LAYOUT_FULL && k == nullptr)
    return SIP_LQR_ERR_UNSUPPORTED; // only the dedicated kernels read the packed triangles
  if (k == nullptr && want != nullptr && want[0] != 0 && !force_general)
    return SIP_LQR_ERR_UNSUPPORTED; // an explicitly requested variant does not exist
  sip_lqr_plan *p = new (std::nothrow) sip_lqr_plan;
  if (p == nullptr)
    return SIP_LQR_ERR_ALLOC;
  p->dtype = dtype;
  p->layout = layout;
  p->batch = batch;
  p->T = T;
  p->n = n;
  p->m = m;
  p->device = device;
  p->kn = n, p->km = m;
  const char *pad = std::getenv("SIP_LQR_PAD");
  if (k == nullptr && !force_general && dtype == SIP_LQR_F64 && (want == nullptr || want[0] == 0) &&
      !(pad && std::strcmp(pad, "0") == 0)) {
    k = find_embedding_kernel(n, m);
    if (k != nullptr) {
      p->padded = true, p->kn = k->n, p->km = k->m;
      p->name_storage = std::string(k->name) + " embedding (" + std::to_string(n) + "," + std::to_string(m) + ")";
    }
  }
  p->kernel_name = p->padded ? p->name_storage.c_str()
                   : k       ? k->name
                             : (dtype == SIP_LQR_F32 ? "tree_generic(chain layout)/f32"
                                                     : "tree_generic(chain layout)/f64");
  p->ws_slot = k ? k->ws_slot : 0;
  p->launch_fs = k ? k->launch_fs : nullptr;
  p->launch_mrhs = (k != nullptr && !p->padded) ? k->launch_mrhs : nullptr;
  p->solve_only = k != nullptr && k->dtype == SIP_LQR_F64 && k->n <= 16; // qw16 only (mt16 re-runs the sweep)
  const char *split = std::getenv("SIP_LQR_SPLIT");
  p->split_on_fused = p->launch_fs != nullptr && (layout != SIP_LQR_LAYOUT_FULL || !(split && std::strcmp(split, "general") == 0));
  init_generic(p);
  // Device tables of the general engine: uploaded here, never lazily (include/sip_lqr_amd.h
  // promises that the compute entry points neither allocate nor synchronise).  A host without any
  // HIP device still gets a plan for the host-side entry points.
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess)
    ndev = 0;
  if (ndev > 0) {
    if (device < 0 || device >= ndev) {
      delete p;
      return SIP_LQR_ERR_INVALID_ARGUMENT;
    }
    const hipError_t e = p->gen.upload(device); // restores the caller's current device
    if (e != hipSuccess) {
      std::fprintf(stderr, "sip_lqr_plan_create: HIP error: %s\n", hipGetErrorString(e));
      delete p;
      return SIP_LQR_ERR_HIP;
    }
    p->on_device = true;
  } else {
    (void)hipGetLastError(); // hipErrorNoDevice is not sticky for the caller
  }
  *plan = p;
  return SIP_LQR_OK;
}

void sip_lqr_plan_destroy(sip_lqr_plan *plan) { delete plan; }

int64_t sip_lqr_plan_batch(const sip_lqr_plan *p) { return p ? p->batch : 0; }
size_t sip_lqr_scalar_bytes(const sip_lqr_plan *p) { return p ? scalar_size(p) : 0; }
size_t sip_lqr_mats_len(const sip_lqr_plan *p) {
  const size_t n = p->n, m = p->m, T = p->T;
  const bool sym = p->layout == SIP_LQR_LAYOUT_SYMMETRIC;
  const size_t qlen = sym ? n * (n + 1) / 2 : n * n, rlen = sym ? m * (m + 1) / 2 : m * m;
  return (T + 1) * (qlen + n) + T * (n * n + 2 * n * m + rlen);
}
size_t sip_lqr_vecs_len(const sip_lqr_plan *p) {
  const size_t n = p->n, m = p->m, T = p->T;
  return (T + 1) * 2 * n + T * m;
}
size_t sip_lqr_gains_len(const sip_lqr_plan *p) {
  const size_t n = p->n, m = p->m, T = p->T;
  return T * (m * n + m);
}
size_t sip_lqr_mats_bytes(const sip_lqr_plan *p) {
  return (size_t)p->batch * sip_lqr_mats_len(p) * scalar_size(p);
}
size_t sip_lqr_vecs_bytes(const sip_lqr_plan *p) {
  return (size_t)p->batch * sip_lqr_vecs_len(p) * scalar_size(p);
}
size_t sip_lqr_sol_bytes(const sip_lqr_plan *p) {
  return sip_lqr_vecs_bytes(p);
}
size_t sip_lqr_gains_bytes(const sip_lqr_plan *p) {
  return (size_t)p->batch * sip_lqr_gains_len(p) * scalar_size(p);
}
size_t sip_lqr_status_bytes(const sip_lqr_plan *p) {
  return (size_t)p->batch * sizeof(int32_t);
}
size_t sip_lqr_workspace_bytes(const sip_lqr_plan *p) {
  // one buffer serves the fused kernel (+ the scratch of its split entry points) and the
  // general engine (never both at once): the larger of the two
  return std::max(fused_split_layout(p).total, generic_ws_bytes(p));
}

} // extern "C"

namespace {
template <class S>
void pack_one(const sip_lqr_plan *pl, int64_t p, double *const *Q,
              double *const *M, double *const *R, double *const *q,
              double *const *r, double *const *A, double *const *B,
              double *const *c, double *const *delta, S *mats, S *vecs) {
  const int n = pl->n, m = pl->m, T = pl->T;
  S *mp = mats + (size_t)p * sip_lqr_mats_len(pl);
  S *vp = vecs + (size_t)p * sip_lqr_vecs_len(pl);
  auto put = [](S *&dst, const double *src, int count) {
    for (int i = 0; i < count; ++i)
      dst[i] = (S)src[i];
    dst += count;
  };
  const bool sym = pl->layout == SIP_LQR_LAYOUT_SYMMETRIC;
  auto put_lower = [](S *&dst, const double *src, int dim) { // lower triangle, packed by columns
    for (int col = 0; col < dim; ++col)
      for (int row = col; row < dim; ++row)
        *dst++ = (S)src[row + dim * col];
  };
  for (int i = 0; i <= T; ++i) {
    if (sym)
      put_lower(mp, Q[i], n);
    else
      put(mp, Q[i], n * n);
    put(mp, delta[i], n);
    put(vp, q[i], n);
    put(vp, c[i], n);
    if (i < T) {
      put(mp, A[i], n * n);
      put(mp, B[i], n * m);
      put(mp, M[i], n * m);
      if (sym)
        put_lower(mp, R[i], m);
      else
        put(mp, R[i], m * m);
      put(vp, r[i], m);
    }
  }
}

template <class S>
void unpack_sol(const sip_lqr_plan *pl, int64_t p, const S *sol,
                double *const *x, double *const *u, double *const *y) {
  const int n = pl->n, m = pl->m, T = pl->T;
  const S *sp = sol + (size_t)p * sip_lqr_vecs_len(pl);
  auto get = [](const S *&src, double *dst, int count) {
    for (int i = 0; i < count; ++i)
      dst[i] = (double)src[i];
    src += count;
  };
  for (int i = 0; i <= T; ++i) {
    get(sp, x[i], n);
    get(sp, y[i], n);
    if (i < T)
      get(sp, u[i], m);
  }
}

template <class S>
void unpack_gain(const sip_lqr_plan *pl, int64_t p, const S *gains,
                 double *const *K, double *const *k) {
  const int n = pl->n, m = pl->m, T = pl->T;
  const S *gp = gains + (size_t)p * sip_lqr_gains_len(pl);
  for (int i = 0; i < T; ++i) {
    for (int e = 0; e < m * n; ++e)
      K[i][e] = (double)gp[e];
    gp += m * n;
    for (int e = 0; e < m; ++e)
      k[i][e] = (double)gp[e];
    gp += m;
  }
}
} // namespace

extern "C" {

int sip_lqr_pack_problem(const sip_lqr_plan *plan, int64_t p, double *const *Q,
                         double *const *M, double *const *R, double *const *q,
                         double *const *r, double *const *A, double *const *B,
                         double *const *c, double *const *delta,
                         void *mats_host, void *vecs_host) {
  if (plan == nullptr || p < 0 || p >= plan->batch || !Q || !q || !c ||
      !delta || !mats_host || !vecs_host ||
      (plan->T > 0 && (!M || !R || !r || !A || !B)))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (plan->dtype == SIP_LQR_F32)
    pack_one<float>(plan, p, Q, M, R, q, r, A, B, c, delta, (float *)mats_host,
                    (float *)vecs_host);
  else
    pack_one<double>(plan, p, Q, M, R, q, r, A, B, c, delta,
                     (double *)mats_host, (double *)vecs_host);
  return SIP_LQR_OK;
}

int sip_lqr_unpack_solution(const sip_lqr_plan *plan, int64_t p,
                            const void *sol_host, double *const *x,
                            double *const *u, double *const *y) {
  if (plan == nullptr || p < 0 || p >= plan->batch || !sol_host || !x || !y ||
      (plan->T > 0 && !u))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (plan->dtype == SIP_LQR_F32)
    unpack_sol<float>(plan, p, (const float *)sol_host, x, u, y);
  else
    unpack_sol<double>(plan, p, (const double *)sol_host, x, u, y);
  return SIP_LQR_OK;
}

int sip_lqr_unpack_gains(const sip_lqr_plan *plan, int64_t p,
                         const void *gains_host, double *const *K,
                         double *const *k) {
  if (plan == nullptr || p < 0 || p >= plan->batch || !gains_host ||
      (plan->T > 0 && (!K || !k)))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (plan->dtype == SIP_LQR_F32)
    unpack_gain<float>(plan, p, (const float *)gains_host, K, k);
  else
    unpack_gain<double>(plan, p, (const double *)gains_host, K, k);
  return SIP_LQR_OK;
}

int sip_lqr_factor_solve(const sip_lqr_plan *plan, const void *d_mats,
                         const void *d_vecs, void *d_sol, void *d_gains,
                         int32_t *d_status, void *d_workspace, void *stream) {
  if (plan == nullptr || !d_mats || !d_vecs || !d_sol || !d_status ||
      !d_workspace || (plan->T > 0 && !d_gains))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (require_device(plan) != hipSuccess)
    return report(hipErrorNoDevice, "sip_lqr_factor_solve");
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_lqr_factor_solve(hipSetDevice)");
  if (plan->launch_fs != nullptr)
    return report(run_fused(plan, d_mats, d_vecs, d_sol, d_gains, d_status, d_workspace, s, 0),
                  "sip_lqr_factor_solve");
  // no dedicated kernel for this shape / dtype: general engine, two launches
  hipError_t e = hipSuccess;
  if (e == hipSuccess)
    e = plan->dtype == SIP_LQR_F32
            ? plan->gen.launch_factor<float>(plan->batch, d_mats, d_workspace, d_gains, d_status, s)
            : plan->gen.launch_factor<double>(plan->batch, d_mats, d_workspace, d_gains, d_status, s);
  if (e == hipSuccess)
    e = plan->dtype == SIP_LQR_F32
            ? plan->gen.launch_solve<float>(plan->batch, d_mats, d_vecs, d_workspace, d_gains, d_sol, d_status, s)
            : plan->gen.launch_solve<double>(plan->batch, d_mats, d_vecs, d_workspace, d_gains, d_sol, d_status, s);
  return report(e, "sip_lqr_factor_solve(general)");
}

// The fused sweep with A | B read in place (chain_qw16.hpp, SPLIT; qw16_split.hip).
int sip_lqr_has_split(const sip_lqr_plan *plan) {
  return plan != nullptr && plan->dtype == SIP_LQR_F64 && !plan->padded && plan->launch_fs != nullptr &&
                 plan->kernel_name != nullptr && std::strstr(plan->kernel_name, "staged") != nullptr &&
                 sipamd::find_split_launch(plan->n, plan->m, plan->layout) != nullptr
             ? 1
             : 0;
}

int64_t sip_lqr_split_mats_len(const sip_lqr_plan *plan) {
  if (plan == nullptr)
    return 0;
  const bool sym = plan->layout == SIP_LQR_LAYOUT_SYMMETRIC;
  const long node = (sym ? (long)plan->n * (plan->n + 1) / 2 : (long)plan->n * plan->n) + plan->n;
  return (int64_t)(plan->T + 1) * node +
         (int64_t)plan->T * (sipamd::split_mats_stage(plan->n, plan->m, plan->layout) - node);
}

int sip_lqr_factor_solve_split(const sip_lqr_plan *plan, const void *d_mats, const void *d_ab,
                               int64_t ab_problem_stride, int64_t ab_stage_stride, const void *d_vecs, void *d_sol,
                               void *d_gains, int32_t *d_status, void *d_workspace, void *stream) {
  if (plan == nullptr || !d_mats || !d_vecs || !d_sol || !d_status || !d_workspace ||
      (plan->T > 0 && (!d_gains || !d_ab)) || ab_problem_stride < 0 || ab_stage_stride < 0)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (!sip_lqr_has_split(plan))
    return SIP_LQR_ERR_UNSUPPORTED;
  // The staging code addresses the four problems of a wavefront by 32-bit byte offsets from the first
  // one's A | B: three problem strides plus one stage image must stay below 2^32 bytes.
  const uint64_t ab_block_bytes = ((uint64_t)plan->n * plan->n + (uint64_t)plan->n * plan->m) * 8u;
  if ((uint64_t)ab_problem_stride > (((uint64_t)1 << 32) - 1 - ab_block_bytes) / (3u * 8u))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (require_device(plan) != hipSuccess)
    return report(hipErrorNoDevice, "sip_lqr_factor_solve_split");
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_lqr_factor_solve_split(hipSetDevice)");
  const sipamd::launch_split_t launch = sipamd::find_split_launch(plan->n, plan->m, plan->layout);
  return report(launch(plan->batch, plan->T, d_mats, d_ab, (long)ab_problem_stride, (long)ab_stage_stride, d_vecs,
                       d_sol, d_gains, d_status, d_workspace, (hipStream_t)stream),
                "sip_lqr_factor_solve_split");
}

// Split entry points: always the general engine (its work arena holds the
// reference's factor state W, G_factor, V, F_factor, sqrt_delta(_inv), v).
int sip_lqr_factor(const sip_lqr_plan *plan, const void *d_mats, void *d_gains,
                   int32_t *d_status, void *d_workspace, void *stream) {
  if (plan == nullptr || !d_mats || !d_status || !d_workspace || (plan->T > 0 && !d_gains))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (require_device(plan) != hipSuccess)
    return report(hipErrorNoDevice, "sip_lqr_factor");
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_lqr_factor(hipSetDevice)");
  if (plan->split_on_fused) { // fused sweep on a zero right-hand side: K, statuses
    const FusedSplit f = fused_split_layout(plan);
    char *w = (char *)d_workspace;
    hipError_t e = run_fused(plan, d_mats, nullptr, nullptr, d_gains, d_status, d_workspace, s, 1);
    if (e == hipSuccess)
      e = sipamd::copy_async(w + f.status, d_status, (size_t)plan->batch * sizeof(int32_t), s);
    return report(e, "sip_lqr_factor(fused)");
  }
  hipError_t e = hipSuccess;
  if (e == hipSuccess)
    e = plan->dtype == SIP_LQR_F32
            ? plan->gen.launch_factor<float>(plan->batch, d_mats, d_workspace, d_gains, d_status, s)
            : plan->gen.launch_factor<double>(plan->batch, d_mats, d_workspace, d_gains, d_status, s);
  if (e == hipSuccess) // keep the statuses for sip_lqr_solve
    e = sipamd::copy_async(generic_status(plan, d_workspace), d_status, (size_t)plan->batch * sizeof(int32_t), s);
  return report(e, "sip_lqr_factor");
}

int sip_lqr_solve(const sip_lqr_plan *plan, const void *d_mats, const void *d_vecs, void *d_sol,
                  void *d_gains, void *d_workspace, void *stream) {
  if (plan == nullptr || !d_mats || !d_vecs || !d_sol || !d_workspace || (plan->T > 0 && !d_gains))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (require_device(plan) != hipSuccess)
    return report(hipErrorNoDevice, "sip_lqr_solve");
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_lqr_solve(hipSetDevice)");
  if (plan->split_on_fused) { // the fused sweep again, now with the right-hand side
    const FusedSplit f = fused_split_layout(plan);
    return report(run_fused(plan, d_mats, d_vecs, d_sol, d_gains, (int32_t *)((char *)d_workspace + f.status),
                            d_workspace, s, 2),
                  "sip_lqr_solve(fused)");
  }
  hipError_t e = hipSuccess;
  const int32_t *st = generic_status(plan, d_workspace);
  if (e == hipSuccess)
    e = plan->dtype == SIP_LQR_F32
            ? plan->gen.launch_solve<float>(plan->batch, d_mats, d_vecs, d_workspace, d_gains, d_sol, st, s)
            : plan->gen.launch_solve<double>(plan->batch, d_mats, d_vecs, d_workspace, d_gains, d_sol, st, s);
  return report(e, "sip_lqr_solve");
}

size_t sip_lqr_solve_multi_workspace_bytes(const sip_lqr_plan *plan, int num_rhs) {
  if (plan == nullptr || num_rhs < 1 || plan->launch_mrhs == nullptr || !plan->split_on_fused)
    return 0; // column-by-column path: no extra state
  const int cols = std::min(num_rhs, sipamd::kMrhsColumns);
  return (size_t)plan->batch * (size_t)(plan->T + 1) * (size_t)cols * (size_t)(2 * plan->n + plan->m) * sizeof(double);
}

int sip_lqr_solve_multi(const sip_lqr_plan *plan, const void *d_mats, const void *d_vecs_cols, void *d_sol_cols,
                        int num_rhs, void *d_gains, void *d_workspace, void *d_col_workspace, void *stream) {
  if (plan == nullptr || num_rhs < 0 || !d_mats || !d_workspace || (plan->T > 0 && !d_gains) ||
      (num_rhs > 0 && (!d_vecs_cols || !d_sol_cols)))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const size_t col_bytes = sip_lqr_vecs_bytes(plan);
  if (plan->launch_mrhs == nullptr || !plan->split_on_fused) { // no multi-rhs kernel for this shape
    for (int col = 0; col < num_rhs; ++col) {
      const int rc = sip_lqr_solve(plan, d_mats, (const char *)d_vecs_cols + (size_t)col * col_bytes,
                                   (char *)d_sol_cols + (size_t)col * col_bytes, d_gains, d_workspace, stream);
      if (rc != SIP_LQR_OK)
        return rc;
    }
    return SIP_LQR_OK;
  }
  if (num_rhs > 0 && d_col_workspace == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (require_device(plan) != hipSuccess)
    return report(hipErrorNoDevice, "sip_lqr_solve_multi");
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_lqr_solve_multi(hipSetDevice)");
  const FusedSplit f = fused_split_layout(plan);
  const char *w = (const char *)d_workspace;
  const long stride = (long)(col_bytes / sizeof(double));
  hipError_t e = hipSuccess;
  for (int col0 = 0; col0 < num_rhs && e == hipSuccess; col0 += sipamd::kMrhsColumns) {
    const int nc = std::min(sipamd::kMrhsColumns, num_rhs - col0);
    e = plan->launch_mrhs(plan->batch, plan->T, d_mats, (const char *)d_vecs_cols + (size_t)col0 * col_bytes,
                          (char *)d_sol_cols + (size_t)col0 * col_bytes, d_gains, d_workspace, w + f.gfac,
                          d_col_workspace, (const int32_t *)(w + f.status), nc, stride, (hipStream_t)stream);
  }
  return report(e, "sip_lqr_solve_multi");
}

const char *sip_lqr_kernel_name(const sip_lqr_plan *plan) {
  return plan ? plan->kernel_name : "";
}

// "sip_lqr_amd <version> (gfx950) SIPLQRSRC=<sha256 of the sources and flags the library was built
// from>": __graft_entry__.build_hip() links the stamp in and rebuilds when it differs from the sources
// next to the library; bench.py and smoke() print it, so that a run record names the build it measured.
const char *sip_lqr_version(void) {
  static const std::string v = std::string("sip_lqr_amd 0.2 (gfx950) ") + sip_lqr_build_stamp;
  return v.c_str();
}

} // extern "C"
