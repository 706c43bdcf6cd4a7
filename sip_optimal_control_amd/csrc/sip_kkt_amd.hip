// sip_kkt_amd.hip -- C ABI of the batched Newton-KKT step
// (include/sip_kkt_amd.h) over csrc/kkt_kernels.hpp and the library's own LQR
// entry points (include/sip_lqr_amd.h).
#include "../../include/sip_kkt_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "generic_plan.hpp"
#include "stream_fill.hpp"
#include "kkt_chain_kernels.hpp"
#include "kkt_kernels.hpp"
#include "kkt_theta_kernels.hpp"
#include "kkt_theta_chain_kernels.hpp"

struct sip_kkt_plan {
  int64_t batch = 0;
  int device = 0;
  int E = 0, N = 0, root = 0;
  int input_status = SIP_KKT_INVALID_INPUT;
  std::vector<int> sd, cd, ncd, ngd, ecd, egd, parents, children;
  std::vector<int> voff[7];
  std::vector<long> moff[sipamd::kkt::NUM_BLOCKS];
  int x_dim = 0, y_dim = 0, z_dim = 0;
  long model_len = 0;
  // Riccati back end: packed chain plan or general tree plan
  sip_lqr_plan *chain = nullptr;
  sip_lqr_tree_plan *tree = nullptr;
  long in0_len = 0, in1_len = 0, out_len = 0, gain_len = 0;
  // byte offsets of the regions of d_work
  size_t at_in0 = 0, at_in1 = 0, at_out = 0, at_gain = 0, at_inv = 0, at_reg = 0, at_lqr = 0, work_bytes = 0;
  void *d_ints = nullptr, *d_longs = nullptr;
  sipamd::kkt::Meta meta{};
  bool chain_kernels = false; // uniform chain: arithmetic-offset kernels (kkt_chain_kernels.hpp)
  // the fused step hands the Riccati sweep ddyn_dx | ddyn_du in the model arena instead of copying
  // them into its inputs (sip_lqr_factor_solve_split); SIP_KKT_SPLIT=0 keeps the copy
  bool chain_split = false;
  // > 0: the plan's dimensions are those of the reference's Newton-KKT benchmark family for (n, m) =
  // (family / 100, family % 100) -- the hot chain kernels then run as the instantiation that has every dimension
  // but the horizon as a constant (kkt_chain_kernels.hpp: family_dims); SIP_KKT_FAMILY=0 keeps the generic kernels
  int family = 0;
  // ... and with Q_mod / R_mod as packed lower triangles (SIP_LQR_LAYOUT_SYMMETRIC) where the sweep has that kernel:
  // a second plan of the same shape, used by the fused step only; SIP_KKT_SYM=0 keeps the full squares
  sip_lqr_plan *chain_sym = nullptr;
  int chain_pipe = 0;         // > 0: stages per wavefront of the software-pipelined condensation
  sipamd::kkt::ChainKkt ck{};
  size_t lds_chain_condense = 0, lds_chain_recover = 0, lds_chain_apply = 0;
  // theta (sip_kkt_plan_set_theta)
  int theta_dim = 0;
  std::vector<long> toff[sipamd::kkt::TH_NUM_BLOCKS];
  long theta_len = 0;
  void *d_theta_longs = nullptr;
  sipamd::kkt::ThetaMeta theta_meta{};
  // uniform chains: the fused theta passes of kkt_theta_chain_kernels.hpp (J_theta never assembled);
  // SIP_KKT_THETA_FUSED=0 keeps the generic passes
  bool chain_theta = false;
  sipamd::kkt::ChainTheta ct{};
  size_t lds_theta_rhs = 0, lds_theta_recover = 0, lds_theta_dot = 0;
  bool staged = false; // LDS-staged kernels (false: items too large for LDS, or SIP_KKT_VARIANT=direct)
  size_t lds_condense = 0, lds_rhs = 0, lds_recover = 0;
  std::string name;

  ~sip_kkt_plan() {
    if (chain)
      sip_lqr_plan_destroy(chain);
    if (chain_sym)
      sip_lqr_plan_destroy(chain_sym);
    if (tree)
      sip_lqr_tree_plan_destroy(tree);
    if (d_ints)
      (void)hipFree(d_ints);
    if (d_longs)
      (void)hipFree(d_longs);
    if (d_theta_longs)
      (void)hipFree(d_theta_longs);
  }
};

namespace {

using sipamd::kkt::Meta;

std::vector<int> dims_or_zero(const int *src, int count) {
  return src ? std::vector<int>(src, src + count) : std::vector<int>((size_t)count, 0);
}

// validate_input, types.cpp:68-127 without the in-degree / reachability part,
// which the LQR traversal (lqr.cpp:563-631) latches.
bool validate(const sip_kkt_plan &p, bool have_topology) {
  for (int i = 0; i < p.N; ++i)
    if (p.sd[i] < 0 || p.ncd[i] < 0 || p.ngd[i] < 0)
      return false;
  for (int e = 0; e < p.E; ++e)
    if (p.cd[e] < 0 || p.ecd[e] < 0 || p.egd[e] < 0)
      return false;
  if (!have_topology || p.root < 0 || p.root >= p.N)
    return false;
  for (int e = 0; e < p.E; ++e) {
    const int a = p.parents[e], c = p.children[e];
    if (a < 0 || a >= p.N || c < 0 || c >= p.N || a == c)
      return false;
  }
  return true;
}

bool is_uniform_chain(const sip_kkt_plan &p) {
  if (p.E < 1 || p.root != 0 || p.sd[0] < 1 || p.cd[0] < 1)
    return false;
  for (int e = 0; e < p.E; ++e)
    if (p.parents[e] != e || p.children[e] != e + 1 || p.cd[e] != p.cd[0])
      return false;
  for (int i = 0; i < p.N; ++i)
    if (p.sd[i] != p.sd[0])
      return false;
  return true;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

int report(hipError_t e, const char *what) {
  if (e == hipSuccess)
    return SIP_LQR_OK;
  std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
  return SIP_LQR_ERR_HIP;
}

struct Regions {
  double *in0, *in1, *out, *gain, *inv;
  int *reg;
  void *lqr;
};
Regions regions(const sip_kkt_plan *p, void *work) {
  char *w = (char *)work;
  Regions r;
  r.in0 = (double *)(w + p->at_in0), r.in1 = (double *)(w + p->at_in1);
  r.out = (double *)(w + p->at_out), r.gain = (double *)(w + p->at_gain);
  r.inv = (double *)(w + p->at_inv), r.reg = (int *)(w + p->at_reg), r.lqr = w + p->at_lqr;
  return r;
}

unsigned item_grid(const sip_kkt_plan *p) { return (unsigned)(p->batch * (p->N + p->E)); }
unsigned node_grid(const sip_kkt_plan *p) { return (unsigned)(p->batch * p->N); }

// Uniform chain whose constraint dimensions are uniform too (interior nodes, terminal node, edges).
bool uniform_constraints(const sip_kkt_plan &p) {
  for (int i = 1; i < p.E; ++i)
    if (p.ncd[i] != p.ncd[0] || p.ngd[i] != p.ngd[0])
      return false;
  for (int e = 1; e < p.E; ++e)
    if (p.ecd[e] != p.ecd[0] || p.egd[e] != p.egd[0])
      return false;
  return p.sd[0] <= 32 && p.cd[0] <= 32;
}
int even(int v) { return (v + 1) / 2 * 2; }

// Calls f(integral_constant<FN>, integral_constant<FM>) for the plan's benchmark-family shape (family = 100 n + m),
// f(0, 0) -- the generic kernels -- otherwise.
template <class F> void family_dispatch(const int family, F &&f) {
  auto at = [&](auto fn, auto fm) { f(fn, fm); };
#define SIP_KKT_FAMILY_CASE(FN, FM)                                                      \
  case 100 * FN + FM:                                                                    \
    at(std::integral_constant<int, FN>{}, std::integral_constant<int, FM>{});           \
    break;
  switch (family) {
    SIP_KKT_FAMILY_CASE(4, 1) SIP_KKT_FAMILY_CASE(4, 2) SIP_KKT_FAMILY_CASE(4, 3) SIP_KKT_FAMILY_CASE(4, 4)
    SIP_KKT_FAMILY_CASE(6, 1) SIP_KKT_FAMILY_CASE(6, 2) SIP_KKT_FAMILY_CASE(6, 3) SIP_KKT_FAMILY_CASE(6, 4)
    SIP_KKT_FAMILY_CASE(8, 1) SIP_KKT_FAMILY_CASE(8, 2) SIP_KKT_FAMILY_CASE(8, 3) SIP_KKT_FAMILY_CASE(8, 4)
    SIP_KKT_FAMILY_CASE(12, 1) SIP_KKT_FAMILY_CASE(12, 2) SIP_KKT_FAMILY_CASE(12, 3) SIP_KKT_FAMILY_CASE(12, 4)
  default:
    at(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  }
#undef SIP_KKT_FAMILY_CASE
}

// b != nullptr (fused factor+solve on the staged kernels): also builds q_mod, r_mod, c_mod.
// split (chain kernels only): mats for sip_lqr_factor_solve_split -- no A | B in it.
hipError_t launch_condense(const sip_kkt_plan *p, const Regions &r, const double *model, const double *w,
                           const double *r1, const double *r2, const double *r3, const double *b,
                           hipStream_t s, const bool split = false, const bool sym = false) {
  sipamd::kkt::ChainKkt ck = p->ck;
  if (split) {
    const int n = ck.n, m = ck.m;
    const int qlen = sym ? n * (n + 1) / 2 : n * n, rlen = sym ? m * (m + 1) / 2 : m * m;
    ck.split = 1, ck.sym = sym ? 1 : 0;
    ck.mats_stage = (qlen + n) + (n * m + rlen);
    ck.mats_len = (long)(ck.T + 1) * (qlen + n) + (long)ck.T * (n * m + rlen);
  }
  hipError_t e = sipamd::zero_async(r.reg, (size_t)p->batch * sizeof(int), s); // a kernel: stream_fill.hpp
  if (e != hipSuccess)
    return e;
  const long per = (long)p->y_dim + p->z_dim;
  if (per > 0)
    hipLaunchKernelGGL(sipamd::kkt::weights_kernel, dim3((unsigned)((p->batch * per + 255) / 256)), dim3(256), 0, s,
                       p->meta, w, r2, r3, r.inv, r.reg, (long)p->batch);
  // the pipelined kernel copies 16-byte pieces: the arena itself has to be 16-byte aligned
  const bool pipe = p->chain_kernels && p->chain_pipe > 0 && ((uintptr_t)model & 7) == 0;
  // the last item of the arena, if of odd length: its last 16-byte piece would reach 8 bytes past the caller's array
  const bool odd_tail = pipe && ((ck.n * ck.n + (ck.cT + ck.gT) * ck.n) & 1);
  const long pipe_batch = odd_tail ? (long)p->batch - 1 : (long)p->batch;
  const unsigned pipe_grid = pipe ? (unsigned)((pipe_batch * p->N + p->chain_pipe - 1) / p->chain_pipe) : 0u;
  // LDS of the chain kernels: image | weights | weighted rows | r1 slice | the outgoing block of mats -- of THIS
  // layout (split / packed blocks are smaller: more workgroups per CU)
  const size_t lds_chain = p->chain_kernels ? sizeof(double) * ((size_t)ck.lds_item + 2 * (size_t)ck.lds_rows +
                                                                 (size_t)even(ck.n + ck.m) + (size_t)ck.mats_stage)
                                            : 0;
  // one-stage kernel over `nb` problems starting at problem `q0`
  auto one_stage = [&](const long q0, const long nb) {
    const long kkt_len = (long)p->x_dim + p->y_dim + p->z_dim;
    const double *mq = model + q0 * ck.model_len, *r1q = r1 + q0 * p->x_dim, *invq = r.inv + q0 * ((long)p->y_dim + p->z_dim);
    double *in0q = r.in0 + q0 * ck.mats_len;
    if (b != nullptr) {
      const double *bq = b + q0 * kkt_len;
      double *in1q = r.in1 + q0 * ck.vecs_len;
      family_dispatch(p->family, [&](auto fn, auto fm) {
        hipLaunchKernelGGL((sipamd::kkt::condense_chain_kernel<true, true, decltype(fn)::value, decltype(fm)::value>),
                           dim3((unsigned)(nb * p->N)), dim3(sipamd::kkt::TPB), lds_chain, s, ck, mq, r1q, invq, in0q, bq,
                           in1q, nb);
      });
    } else {
      family_dispatch(p->family, [&](auto fn, auto fm) {
        hipLaunchKernelGGL((sipamd::kkt::condense_chain_kernel<false, true, decltype(fn)::value, decltype(fm)::value>),
                           dim3((unsigned)(nb * p->N)), dim3(sipamd::kkt::TPB), lds_chain, s, ck, mq, r1q, invq, in0q,
                           (const double *)nullptr, (double *)nullptr, nb);
      });
    }
  };
  if (pipe) {
    if (pipe_batch > 0 && b != nullptr)
      family_dispatch(p->family, [&](auto fn, auto fm) {
        hipLaunchKernelGGL((sipamd::kkt::condense_chain_pipe_kernel<true, decltype(fn)::value, decltype(fm)::value>),
                           dim3(pipe_grid), dim3(sipamd::kkt::TPB), lds_chain, s, ck, model, r1, r.inv, r.in0, b,
                           r.in1, pipe_batch, p->chain_pipe);
      });
    else if (pipe_batch > 0)
      family_dispatch(p->family, [&](auto fn, auto fm) {
        hipLaunchKernelGGL((sipamd::kkt::condense_chain_pipe_kernel<false, decltype(fn)::value, decltype(fm)::value>),
                           dim3(pipe_grid), dim3(sipamd::kkt::TPB), lds_chain, s, ck, model, r1, r.inv, r.in0,
                           (const double *)nullptr, (double *)nullptr, pipe_batch, p->chain_pipe);
      });
    if (odd_tail)
      one_stage((long)p->batch - 1, 1);
  } else if (p->chain_kernels) { // (SIP_KKT_PIPE=0, or an item longer than one pass of the pipelined walk)
    one_stage(0, (long)p->batch);
  }
  else if (p->staged && b != nullptr) // condensation and right-hand side from one staging of the model
    hipLaunchKernelGGL(sipamd::kkt::condense_staged_kernel<true>, dim3(node_grid(p)), dim3(sipamd::kkt::TPB),
                       p->lds_condense, s, p->meta, model, r1, r.inv, r.in0, b, r.in1, (long)p->batch);
  else if (p->staged)
    hipLaunchKernelGGL(sipamd::kkt::condense_staged_kernel<false>, dim3(node_grid(p)), dim3(sipamd::kkt::TPB),
                       p->lds_condense, s, p->meta, model, r1, r.inv, r.in0, (const double *)nullptr,
                       (double *)nullptr, (long)p->batch);
  else
    hipLaunchKernelGGL(sipamd::kkt::condense_kernel, dim3(item_grid(p)), dim3(sipamd::kkt::TPB), 0, s, p->meta,
                       model, r1, r.inv, r.in0, (long)p->batch);
  return hipGetLastError();
}

hipError_t launch_merge(const sip_kkt_plan *p, const Regions &r, int32_t *status, hipStream_t s) {
  hipLaunchKernelGGL(sipamd::kkt::merge_status_kernel, dim3((unsigned)((p->batch + 255) / 256)), dim3(256), 0, s,
                     r.reg, status, (long)p->batch, (int)SIP_KKT_NONPOSITIVE_REGULARIZATION);
  return hipGetLastError();
}

hipError_t launch_rhs(const sip_kkt_plan *p, const Regions &r, const double *model, const double *b,
                      const int32_t *status, hipStream_t s) {
  if (p->chain_kernels)
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::condense_chain_kernel<true, false, decltype(fn)::value, decltype(fm)::value>),
                         dim3(node_grid(p)), dim3(sipamd::kkt::TPB),
                         // (rhs only: the constraint Jacobians | weights | weighted rows)
                         sizeof(double) * ((size_t)p->ck.lds_tail + 2 * (size_t)p->ck.lds_rows), s, p->ck, model,
                         (const double *)nullptr, r.inv, r.in0, b, r.in1, (long)p->batch, status);
    });
  else if (p->staged)
    hipLaunchKernelGGL(sipamd::kkt::rhs_staged_kernel, dim3(node_grid(p)), dim3(sipamd::kkt::TPB), p->lds_rhs, s,
                       p->meta, model, b, r.inv, r.in1, status, (long)p->batch);
  else
    hipLaunchKernelGGL(sipamd::kkt::rhs_kernel, dim3(item_grid(p)), dim3(sipamd::kkt::TPB), 0, s, p->meta, model,
                       b, r.inv, r.in1, status, (long)p->batch);
  return hipGetLastError();
}

hipError_t launch_recover(const sip_kkt_plan *p, const Regions &r, const double *model, const double *b,
                          double *sol, const int32_t *status, hipStream_t s) {
  if (p->chain_kernels)
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::recover_chain_kernel<false, decltype(fn)::value, decltype(fm)::value>),
                         dim3(node_grid(p)), dim3(sipamd::kkt::TPB), p->lds_chain_recover, s, p->ck, model, b, r.inv, r.out,
                         sol, status, (long)p->batch);
    });
  else if (p->staged)
    hipLaunchKernelGGL(sipamd::kkt::recover_staged_kernel, dim3(node_grid(p)), dim3(sipamd::kkt::TPB),
                       p->lds_recover, s, p->meta, model, b, r.inv, r.out, sol, status, (long)p->batch);
  else
    hipLaunchKernelGGL(sipamd::kkt::recover_kernel, dim3(item_grid(p)), dim3(sipamd::kkt::TPB), 0, s, p->meta,
                       model, b, r.inv, r.out, sol, status, (long)p->batch);
  return hipGetLastError();
}

// [x | theta | y | z] vectors of the add_Kx_to_y entry points as an ApplyIO (all blocks).
sipamd::kkt::ApplyIO kkt_vector_io(const sip_kkt_plan *p, int theta_dim, const double *d_x, double *d_y) {
  const long xt = (long)p->x_dim + theta_dim, full = xt + p->y_dim + p->z_dim;
  sipamd::kkt::ApplyIO io;
  io.x_x = d_x, io.x_y = d_x + xt, io.x_z = d_x + xt + p->y_dim;
  io.y_x = d_y, io.y_y = d_y + xt, io.y_z = d_y + xt + p->y_dim;
  io.sx = io.sy = io.sz = full;
  io.parts = sipamd::kkt::AP_ALL;
  return io;
}

// y += (selected blocks of K) x.  d_theta != nullptr: x-space vectors carry theta behind the
// stagewise x and the theta sections of the operators are added (apply_theta_kernel).
int apply_blocks(const sip_kkt_plan *p, const double *d_model, const double *d_theta, const double *d_w,
                 const double *d_r1, const double *d_r2, const double *d_r3, const sipamd::kkt::ApplyIO &io,
                 hipStream_t s, const char *what) {
  sipamd::DeviceGuard on_device(p->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return report(on_device.err, what);
  const int th = d_theta != nullptr ? p->theta_dim : 0;
  if (p->chain_kernels) {
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::apply_chain_kernel<decltype(fn)::value, decltype(fm)::value>), dim3(node_grid(p)),
                         dim3(sipamd::kkt::TPB), p->lds_chain_apply, s, p->ck, th, d_model, d_w, d_r1, d_r2, d_r3, io,
                         (long)p->batch);
    });
  } else {
    sipamd::kkt::Meta wide = p->meta; // x-space = [stagewise x | theta]
    wide.theta_dim = th;
    hipLaunchKernelGGL(sipamd::kkt::apply_kernel, dim3(item_grid(p)), dim3(sipamd::kkt::TPB), 0, s, wide, d_model,
                       d_w, d_r1, d_r2, d_r3, io, (long)p->batch);
  }
  if (th > 0 && p->chain_theta) {
    // W wavefronts per problem, as many as 64 KiB of LDS hold (kkt_theta_chain_kernels.hpp)
    const size_t pe = ((size_t)th + 1) & ~(size_t)1;
    const size_t vlen = ((size_t)(2 * p->ck.n + p->ck.m) + 1) & ~(size_t)1;
    const size_t per_wave = sizeof(double) * ((size_t)p->ct.lds_item + vlen + p->ck.lds_rows + 2 * pe);
    int waves = (int)std::min<size_t>(8, (64 * 1024) / per_wave);
    waves = std::max(1, std::min(waves, p->N));
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::apply_theta_chain_kernel<decltype(fn)::value, decltype(fm)::value>),
                         dim3((unsigned)p->batch), dim3(64 * waves), per_wave * waves, s, p->ck, p->ct, d_theta, d_r1, io,
                         (long)p->batch);
    });
  } else if (th > 0)
    hipLaunchKernelGGL(sipamd::kkt::apply_theta_kernel, dim3((unsigned)p->batch), dim3(sipamd::kkt::TPB), 0, s,
                       p->meta, p->theta_meta, d_theta, d_r1, io, (long)p->batch);
  return report(hipGetLastError(), what);
}

// One of the five block operators: xs / ys = vector space of x / y (0 x-space, 1 y-space, 2 z-space).
int block_op(const sip_kkt_plan *p, const double *d_model, const double *d_theta, int part, int xs, int ys,
             const double *d_x, double *d_y, hipStream_t s, const char *what) {
  if (p == nullptr || p->input_status != SIP_KKT_SUCCESS)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const int th = d_theta != nullptr ? p->theta_dim : 0;
  const long len[3] = {(long)p->x_dim + th, (long)p->y_dim, (long)p->z_dim};
  if (len[xs] == 0 || len[ys] == 0)
    return SIP_LQR_OK; // an empty space: nothing to add
  if ((!d_model && p->model_len > 0) || !d_x || !d_y)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sipamd::kkt::ApplyIO io{};
  io.sx = len[0], io.sy = len[1], io.sz = len[2];
  (xs == 0 ? io.x_x : xs == 1 ? io.x_y : io.x_z) = d_x;
  (ys == 0 ? io.y_x : ys == 1 ? io.y_y : io.y_z) = d_y;
  io.parts = part;
  return apply_blocks(p, d_model, d_theta, nullptr, nullptr, nullptr, nullptr, io, s, what);
}

int fill_status(const sip_kkt_plan *p, int32_t *d_status, hipStream_t s) {
  std::vector<int32_t> st((size_t)p->batch, p->input_status);
  if (hipMemcpyAsync(d_status, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return SIP_LQR_ERR_HIP;
  return SIP_LQR_OK;
}

} // namespace

extern "C" {

int sip_kkt_plan_create(int64_t batch, int num_edges, int root, const int *edge_parents,
                        const int *edge_children, const int *state_dims, const int *control_dims,
                        const int *node_c_dims, const int *node_g_dims, const int *edge_c_dims,
                        const int *edge_g_dims, int device, sip_kkt_plan **out) {
  if (out == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (batch < 1 || num_edges < 0 || state_dims == nullptr || (num_edges > 0 && control_dims == nullptr))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sip_kkt_plan *p = new (std::nothrow) sip_kkt_plan;
  if (p == nullptr)
    return SIP_LQR_ERR_ALLOC;
  const int E = num_edges, N = E + 1;
  p->batch = batch, p->device = device, p->E = E, p->N = N, p->root = root;
  p->sd.assign(state_dims, state_dims + N);
  p->cd = dims_or_zero(control_dims, E);
  p->ncd = dims_or_zero(node_c_dims, N), p->ngd = dims_or_zero(node_g_dims, N);
  p->ecd = dims_or_zero(edge_c_dims, E), p->egd = dims_or_zero(edge_g_dims, E);
  const bool have_topology = E == 0 || (edge_parents != nullptr && edge_children != nullptr);
  p->parents = dims_or_zero(edge_parents, E), p->children = dims_or_zero(edge_children, E);
  for (auto &v : p->voff)
    v.assign(N, 0);
  for (auto &v : p->moff)
    v.assign(N, 0);
  *out = p;
  if (!validate(*p, have_topology)) {
    p->name = "invalid input";
    return SIP_LQR_OK; // latched (helpers.cpp:24-26)
  }

  // model arena: node i, then edge i
  using namespace sipamd::kkt;
  long at = 0;
  for (int i = 0; i < N; ++i) {
    const long n = p->sd[i];
    p->moff[N_Q][i] = at, at += n * n;
    p->moff[N_JC][i] = at, at += p->ncd[i] * n;
    p->moff[N_JG][i] = at, at += p->ngd[i] * n;
    if (i < E) {
      const int e = i;
      const long np = p->sd[p->parents[e]], nc = p->sd[p->children[e]], m = p->cd[e];
      p->moff[E_Q][e] = at, at += np * np;
      p->moff[E_M][e] = at, at += np * m;
      p->moff[E_R][e] = at, at += m * m;
      p->moff[E_A][e] = at, at += nc * np;
      p->moff[E_B][e] = at, at += nc * m;
      p->moff[E_JXC][e] = at, at += p->ecd[e] * np;
      p->moff[E_JUC][e] = at, at += p->ecd[e] * m;
      p->moff[E_JXG][e] = at, at += p->egd[e] * np;
      p->moff[E_JUG][e] = at, at += p->egd[e] * m;
    }
  }
  p->model_len = at;

  // populate_workspace_metadata, types.cpp:24-64
  int xo = 0, yo = 0, zo = 0;
  for (int i = 0; i < N; ++i) {
    p->voff[SIP_KKT_X_STATE][i] = xo;
    if (i < E) {
      xo += p->sd[i];
      p->voff[SIP_KKT_X_CONTROL][i] = xo;
      xo += p->cd[i];
    }
  }
  xo = p->sd[E];
  for (int e = 0; e < E; ++e)
    xo += p->sd[e] + p->cd[e];
  std::vector<int> y_is_dyn;
  for (int i = 0; i < N; ++i) {
    p->voff[SIP_KKT_Y_DYN][i] = yo, yo += p->sd[i];
    p->voff[SIP_KKT_Y_NODE_C][i] = yo, yo += p->ncd[i];
    y_is_dyn.insert(y_is_dyn.end(), (size_t)p->sd[i], 1);
    y_is_dyn.insert(y_is_dyn.end(), (size_t)p->ncd[i], 0);
  }
  for (int e = 0; e < E; ++e) {
    p->voff[SIP_KKT_Y_EDGE_C][e] = yo, yo += p->ecd[e];
    y_is_dyn.insert(y_is_dyn.end(), (size_t)p->ecd[e], 0);
  }
  for (int i = 0; i < N; ++i)
    p->voff[SIP_KKT_Z_NODE][i] = zo, zo += p->ngd[i];
  for (int e = 0; e < E; ++e)
    p->voff[SIP_KKT_Z_EDGE][e] = zo, zo += p->egd[e];
  p->x_dim = xo, p->y_dim = yo, p->z_dim = zo;

  // Riccati back end + the offsets of its arenas
  sipamd::GenericPlan g; // host tables only
  g.set_shape(E, root, p->sd.data(), p->cd.data());
  g.parents = p->parents, g.children = p->children;
  size_t lqr_bytes = 0;
  int rc = SIP_LQR_OK;
  if (is_uniform_chain(*p)) {
    rc = sip_lqr_plan_create(SIP_LQR_F64, batch, E, p->sd[0], p->cd[0], device, &p->chain);
    if (rc == SIP_LQR_OK) {
      g.layout_chain(p->sd[0], p->cd[0], E);
      p->in0_len = g.in0_len, p->in1_len = g.in1_len, p->out_len = g.out_len, p->gain_len = g.gain_len;
      lqr_bytes = sip_lqr_workspace_bytes(p->chain);
      p->name = std::string("chain:") + sip_lqr_kernel_name(p->chain);
    }
  } else {
    const int none = 0; // an edge-less tree still needs non-null edge arrays (lqr.cpp:566-569)
    rc = sip_lqr_tree_plan_create(batch, E, root, E ? p->parents.data() : &none, E ? p->children.data() : &none,
                                  p->sd.data(), E ? p->cd.data() : &none, device, &p->tree);
    if (rc == SIP_LQR_OK && sip_lqr_tree_topology_status(p->tree) != SIP_LQR_SUCCESS) {
      p->input_status = sip_lqr_tree_topology_status(p->tree); // INVALID_TOPOLOGY, latched
      p->name = "invalid topology";
      return SIP_LQR_OK;
    }
    if (rc == SIP_LQR_OK) {
      g.layout_tree_native();
      p->in0_len = g.in0_len, p->in1_len = 0, p->out_len = g.out_len, p->gain_len = 0;
      lqr_bytes = sip_lqr_tree_work_len(p->tree) * sizeof(double) * (size_t)batch;
      p->name = "tree:general";
    }
  }
  if (rc != SIP_LQR_OK) {
    delete p;
    *out = nullptr;
    return rc;
  }
  size_t cur = 0;
  const size_t B = (size_t)batch * sizeof(double);
  p->at_in0 = cur, cur = align256(cur + B * (size_t)p->in0_len);
  p->at_in1 = p->chain ? cur : p->at_in0; // the tree arena holds q, r, c too
  if (p->chain)
    cur = align256(cur + B * (size_t)p->in1_len);
  p->at_out = cur, cur = align256(cur + B * (size_t)p->out_len);
  p->at_gain = cur, cur = align256(cur + B * (size_t)p->gain_len);
  p->at_inv = cur, cur = align256(cur + B * (size_t)(p->y_dim + p->z_dim));
  p->at_reg = cur, cur = align256(cur + (size_t)batch * sizeof(int));
  p->at_lqr = cur, cur = align256(cur + lqr_bytes);
  p->work_bytes = cur;
  if (!p->chain)
    p->in1_len = p->in0_len; // per-problem stride of the arena rhs_kernel writes

  // incoming edge of every node (the root has none); child CSR as the LQR compiles it
  std::vector<int> in_edge(N, -1), child_offsets(N + 1, 0), child_edges(E, 0);
  for (int e = 0; e < E; ++e)
    in_edge[p->children[e]] = e, ++child_offsets[p->parents[e] + 1];
  for (int i = 0; i < N; ++i)
    child_offsets[i + 1] += child_offsets[i];
  {
    std::vector<int> cursor(child_offsets.begin(), child_offsets.end() - 1);
    for (int e = 0; e < E; ++e)
      child_edges[cursor[p->parents[e]]++] = e; // stable in the edge index (lqr.cpp:588-598)
  }

  // LDS plan of the staged kernels
  int lds_q = 0, lds_item = 0, lds_tail = 0, lds_rows = 1;
  for (int i = 0; i < N; ++i) {
    const int n = p->sd[i], cg = p->ncd[i] + p->ngd[i];
    lds_q = std::max(lds_q, n * n), lds_item = std::max(lds_item, n * n + cg * n);
    lds_tail = std::max(lds_tail, cg * n), lds_rows = std::max(lds_rows, std::max(cg, n));
  }
  for (int e = 0; e < E; ++e) {
    const int n = p->sd[p->parents[e]], nc = p->sd[p->children[e]], m = p->cd[e], cg = p->ecd[e] + p->egd[e];
    lds_item = std::max(lds_item, n * n + n * m + m * m + nc * n + nc * m + cg * (n + m));
    lds_tail = std::max(lds_tail, cg * (n + m)), lds_rows = std::max(lds_rows, std::max(cg, m));
  }
  p->lds_condense = sizeof(double) * ((size_t)lds_q + lds_item + lds_tail + lds_rows);
  p->lds_rhs = sizeof(double) * ((size_t)lds_tail + lds_rows);
  p->lds_recover = sizeof(double) * ((size_t)lds_tail + 2 * (size_t)lds_rows);
  const char *variant = std::getenv("SIP_KKT_VARIANT");
  p->staged = p->lds_condense <= 48 * 1024 && !(variant && std::strcmp(variant, "direct") == 0);
  if (p->chain != nullptr && p->staged && uniform_constraints(*p)) {
    sipamd::kkt::ChainKkt &ck = p->ck;
    ck.n = p->sd[0], ck.m = p->cd[0], ck.T = E;
    ck.cn = E > 1 ? p->ncd[0] : 0, ck.gn = E > 1 ? p->ngd[0] : 0;
    if (E == 1) // one interior node only: its dimensions are the "interior" ones
      ck.cn = p->ncd[0], ck.gn = p->ngd[0];
    ck.cT = p->ncd[E], ck.gT = p->ngd[E], ck.ce = p->ecd[0], ck.ge = p->egd[0];
    const int n = ck.n, mm = ck.m;
    ck.node_len = n * n + (ck.cn + ck.gn) * n;
    ck.edge_len = 2 * n * n + 2 * n * mm + mm * mm + (ck.ce + ck.ge) * (n + mm);
    ck.model_len = p->model_len, ck.x_dim = p->x_dim, ck.y_dim = p->y_dim, ck.z_dim = p->z_dim;
    ck.mats_stage = (n * n + n) + (n * n + 2 * n * mm + mm * mm), ck.vecs_stage = 2 * n + mm;
    ck.mats_len = p->in0_len, ck.vecs_len = p->in1_len;
    ck.split = 0, ck.sym = 0;
    const int cgn = std::max(ck.cn + ck.gn, ck.cT + ck.gT), cge = ck.ce + ck.ge;
    ck.lds_item = even(n * n + cgn * n + ck.edge_len);
    ck.lds_tail = even(cgn * n + cge * (n + mm)); // recover: every Jacobian of a stage
    ck.lds_rows = even(cgn + cge);                // condense: weights | weighted rhs rows
    // model image | weights | weighted rhs rows | r1 of the stage | the stage block of mats on its way out
    p->lds_chain_condense = sizeof(double) * ((size_t)ck.lds_item + 2 * (size_t)ck.lds_rows + even(n + mm) + ck.mats_stage);
    p->lds_chain_recover = sizeof(double) * ((size_t)ck.lds_tail + n + mm);
    p->lds_chain_apply = sizeof(double) * ((size_t)ck.lds_item + 3 * n + mm + (size_t)ck.lds_rows);
    // lane maps of the chain kernels: state rows on lanes 0..n-1, control rows on lanes 32..32+m-1
    p->chain_kernels = p->lds_chain_condense <= 48 * 1024 && n <= 32 && mm <= 32 &&
                       !(variant && std::strcmp(variant, "tables") == 0);
  }
  if (p->chain_kernels) {
    // software-pipelined condensation (SIP_KKT_PIPE = stages per wavefront, 0 = off): the stage
    // image must be one pass of 16-byte pieces at every stage
    const sipamd::kkt::ChainKkt &ck = p->ck;
    const int n = ck.n;
    const int len_mid = ck.node_len + ck.edge_len, len_last = n * n + (ck.cT + ck.gT) * n;
    const char *pe = std::getenv("SIP_KKT_PIPE");
    const int per = pe ? std::atoi(pe) : 6;
    // (any lengths: an image of odd length takes one more piece, whose second half is the first scalar of the item
    // behind it; sources are then only 8-byte aligned, which the vector loads take.  The batch's very last item has
    // nothing behind it: launch_condense gives the last problem to the one-stage kernel when that matters.)
    const bool fits = (len_mid + 1) / 2 <= sipamd::kkt::PIPE_U * sipamd::kkt::TPB &&
                      (len_last + 1) / 2 <= sipamd::kkt::PIPE_U * sipamd::kkt::TPB;
    p->chain_pipe = fits && per > 0 ? per : 0;
    {
      const int fn = ck.n, fm = ck.m, fc = std::max(1, fn / 2), fg = std::max(1, 2 * fm);
      const char *fe = std::getenv("SIP_KKT_FAMILY");
      const bool shape = (fn == 4 || fn == 6 || fn == 8 || fn == 12) && fm >= 1 && fm <= 4;
      if (shape && ck.cn == 0 && ck.gn == 0 && ck.cT == fc && ck.gT == fg && ck.ce == fc && ck.ge == fg &&
          !(fe != nullptr && fe[0] == '0'))
        p->family = 100 * fn + fm;
    }
    const char *se = std::getenv("SIP_KKT_SPLIT");
    p->chain_split = E > 0 && sip_lqr_has_split(p->chain) == 1 && !(se != nullptr && se[0] == '0');
    const char *sy = std::getenv("SIP_KKT_SYM");
    if (p->chain_split && !(sy != nullptr && sy[0] == '0')) {
      if (sip_lqr_plan_create_layout(SIP_LQR_F64, batch, E, p->sd[0], p->cd[0], device, SIP_LQR_LAYOUT_SYMMETRIC,
                                     &p->chain_sym) != SIP_LQR_OK ||
          sip_lqr_has_split(p->chain_sym) != 1 ||
          sip_lqr_workspace_bytes(p->chain_sym) > sip_lqr_workspace_bytes(p->chain)) { // (shares the sweep's scratch)
        sip_lqr_plan_destroy(p->chain_sym);
        p->chain_sym = nullptr;
      }
    }
  }
  p->name += p->chain_kernels ? " + chain condensation" : p->staged ? " + staged condensation" : " + direct condensation";
  if (p->chain_split)
    p->name += p->chain_sym != nullptr ? " (A|B in place, Q|R packed)" : " (A|B in place)";
  if (p->family > 0)
    p->name += " [benchmark-family instantiation]";

  std::vector<int> ints;
  auto pi = [&](const std::vector<int> &v) {
    const size_t where = ints.size();
    ints.insert(ints.end(), v.begin(), v.end());
    return where;
  };
  const size_t a_sd = pi(p->sd), a_cd = pi(p->cd), a_ncd = pi(p->ncd), a_ngd = pi(p->ngd), a_ecd = pi(p->ecd),
               a_egd = pi(p->egd), a_pa = pi(p->parents), a_ch = pi(p->children), a_in = pi(in_edge),
               a_co = pi(child_offsets), a_ce = pi(child_edges), a_dyn = pi(y_is_dyn);
  size_t a_v[7];
  for (int t = 0; t < 7; ++t)
    a_v[t] = pi(p->voff[t]);
  std::vector<long> longs;
  auto pl = [&](const std::vector<long> &v) {
    const size_t where = longs.size();
    longs.insert(longs.end(), v.begin(), v.end());
    return where;
  };
  size_t a_m[NUM_BLOCKS];
  for (int b = 0; b < NUM_BLOCKS; ++b)
    a_m[b] = pl(p->moff[b]);
  const std::vector<long> *lq[12] = {&g.oQ, &g.od, &g.oq, &g.oc, &g.ox, &g.oy, &g.oA, &g.oB, &g.oM, &g.oR, &g.orr, &g.ou};
  size_t a_l[12];
  for (int t = 0; t < 12; ++t)
    a_l[t] = pl(*lq[t]);

  sipamd::DeviceGuard on_device(device); // the caller's current device is restored on return
  hipError_t he = on_device.err;
  if (he == hipSuccess)
    he = hipMalloc(&p->d_ints, std::max<size_t>(1, ints.size()) * sizeof(int));
  if (he == hipSuccess)
    he = hipMalloc(&p->d_longs, std::max<size_t>(1, longs.size()) * sizeof(long));
  if (he == hipSuccess)
    he = hipMemcpy(p->d_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice);
  if (he == hipSuccess)
    he = hipMemcpy(p->d_longs, longs.data(), longs.size() * sizeof(long), hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    std::fprintf(stderr, "sip_kkt_plan_create: HIP error: %s\n", hipGetErrorString(he));
    delete p;
    *out = nullptr;
    return SIP_LQR_ERR_HIP;
  }
  const int *di = (const int *)p->d_ints;
  const long *dl = (const long *)p->d_longs;
  Meta &m = p->meta;
  m.E = E, m.N = N, m.root = root, m.x_dim = p->x_dim, m.y_dim = p->y_dim, m.z_dim = p->z_dim;
  m.model_len = p->model_len, m.in0_len = p->in0_len, m.in1_len = p->in1_len, m.out_len = p->out_len;
  m.lds_q = lds_q, m.lds_item = lds_item, m.lds_tail = lds_tail, m.lds_rows = lds_rows;
  m.sd = di + a_sd, m.cd = di + a_cd, m.ncd = di + a_ncd, m.ngd = di + a_ngd, m.ecd = di + a_ecd, m.egd = di + a_egd;
  m.parent = di + a_pa, m.child = di + a_ch, m.in_edge = di + a_in;
  m.child_offsets = di + a_co, m.child_edges = di + a_ce, m.y_is_dyn = di + a_dyn;
  m.x_state = di + a_v[0], m.x_control = di + a_v[1], m.y_dyn = di + a_v[2], m.y_node_c = di + a_v[3];
  m.y_edge_c = di + a_v[4], m.z_node = di + a_v[5], m.z_edge = di + a_v[6];
  for (int b = 0; b < NUM_BLOCKS; ++b)
    m.mo[b] = dl + a_m[b];
  const long **dst[12] = {&m.oQ, &m.od, &m.oq, &m.oc, &m.ox, &m.oy, &m.oA, &m.oB, &m.oM, &m.oR, &m.orr, &m.ou};
  for (int t = 0; t < 12; ++t)
    *dst[t] = dl + a_l[t];
  p->input_status = SIP_KKT_SUCCESS;
  return SIP_LQR_OK;
}

void sip_kkt_plan_destroy(sip_kkt_plan *plan) { delete plan; }

int sip_kkt_input_status(const sip_kkt_plan *p) { return p ? p->input_status : SIP_KKT_INVALID_INPUT; }

size_t sip_kkt_len(const sip_kkt_plan *p, int which) {
  if (p == nullptr || p->input_status != SIP_KKT_SUCCESS)
    return 0;
  switch (which) {
  case SIP_KKT_LEN_X: return (size_t)p->x_dim;
  case SIP_KKT_LEN_Y: return (size_t)p->y_dim;
  case SIP_KKT_LEN_Z: return (size_t)p->z_dim;
  case SIP_KKT_LEN_MODEL: return (size_t)p->model_len;
  default: return 0;
  }
}

size_t sip_kkt_model_offset(const sip_kkt_plan *p, int block, int index) {
  if (p == nullptr || p->input_status != SIP_KKT_SUCCESS || block < 0 || block >= SIP_KKT_NUM_BLOCKS ||
      index < 0 || index >= (block <= SIP_KKT_NODE_DG_DX ? p->N : p->E))
    return (size_t)-1;
  return (size_t)p->moff[block][index];
}

size_t sip_kkt_vector_offset(const sip_kkt_plan *p, int table, int index) {
  if (p == nullptr || p->input_status != SIP_KKT_SUCCESS || table < 0 || table > SIP_KKT_Z_EDGE || index < 0)
    return (size_t)-1;
  const bool per_edge = table == SIP_KKT_X_CONTROL || table == SIP_KKT_Y_EDGE_C || table == SIP_KKT_Z_EDGE;
  if (index >= (per_edge ? p->E : p->N))
    return (size_t)-1;
  return (size_t)p->voff[table][index];
}

size_t sip_kkt_work_bytes(const sip_kkt_plan *p) {
  return (p && p->input_status == SIP_KKT_SUCCESS) ? p->work_bytes : 0;
}

const char *sip_kkt_kernel_name(const sip_kkt_plan *p) { return p ? p->name.c_str() : ""; }

int sip_kkt_factor(const sip_kkt_plan *p, const double *d_model, const double *d_w, const double *d_r1,
                   const double *d_r2, const double *d_r3, void *d_work, int32_t *d_status, void *stream) {
  if (p == nullptr || d_status == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sipamd::DeviceGuard on_device(p->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_kkt_factor(hipSetDevice)");
  hipStream_t s = (hipStream_t)stream;
  if (p->input_status != SIP_KKT_SUCCESS)
    return fill_status(p, d_status, s);
  if ((!d_model && p->model_len > 0) || (!d_w && p->z_dim > 0) || (!d_r1 && p->x_dim > 0) ||
      (!d_r2 && p->y_dim > 0) || (!d_r3 && p->z_dim > 0) || !d_work)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const Regions r = regions(p, d_work);
  hipError_t e = launch_condense(p, r, d_model, d_w, d_r1, d_r2, d_r3, nullptr, s);
  if (e != hipSuccess)
    return report(e, "sip_kkt_factor(condense)");
  const int rc = p->chain ? sip_lqr_factor(p->chain, r.in0, r.gain, d_status, r.lqr, s)
                          : sip_lqr_tree_factor(p->tree, r.in0, (double *)r.lqr, d_status, s);
  if (rc != SIP_LQR_OK)
    return rc;
  return report(launch_merge(p, r, d_status, s), "sip_kkt_factor(status)");
}

int sip_kkt_solve(const sip_kkt_plan *p, const double *d_model, const double *d_b, double *d_sol, void *d_work,
                  const int32_t *d_status, void *stream) {
  if (p == nullptr || d_status == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (p->input_status != SIP_KKT_SUCCESS)
    return SIP_LQR_OK; // nothing was factored; solve() after a false factor() is undefined in the reference
  const long kkt = (long)p->x_dim + p->y_dim + p->z_dim;
  if ((!d_model && p->model_len > 0) || (kkt > 0 && (!d_b || !d_sol)) || !d_work)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sipamd::DeviceGuard on_device(p->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_kkt_solve(hipSetDevice)");
  hipStream_t s = (hipStream_t)stream;
  const Regions r = regions(p, d_work);
  hipError_t e = launch_rhs(p, r, d_model, d_b, d_status, s);
  if (e != hipSuccess)
    return report(e, "sip_kkt_solve(rhs)");
  const int rc = p->chain ? sip_lqr_solve(p->chain, r.in0, r.in1, r.out, r.gain, r.lqr, s)
                          : sip_lqr_tree_solve(p->tree, r.in0, (double *)r.lqr, r.out, d_status, s);
  if (rc != SIP_LQR_OK)
    return rc;
  return report(launch_recover(p, r, d_model, d_b, d_sol, d_status, s), "sip_kkt_solve(recover)");
}

int sip_kkt_factor_solve(const sip_kkt_plan *p, const double *d_model, const double *d_w, const double *d_r1,
                         const double *d_r2, const double *d_r3, const double *d_b, double *d_sol, void *d_work,
                         int32_t *d_status, void *stream) {
  if (p == nullptr || d_status == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (p->input_status != SIP_KKT_SUCCESS || p->chain == nullptr) {
    const int rc = sip_kkt_factor(p, d_model, d_w, d_r1, d_r2, d_r3, d_work, d_status, stream);
    return rc != SIP_LQR_OK ? rc : sip_kkt_solve(p, d_model, d_b, d_sol, d_work, d_status, stream);
  }
  const long kkt = (long)p->x_dim + p->y_dim + p->z_dim;
  if (!d_model || (!d_w && p->z_dim > 0) || !d_r1 || !d_r2 || (!d_r3 && p->z_dim > 0) || !d_work ||
      (kkt > 0 && (!d_b || !d_sol)))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sipamd::DeviceGuard on_device(p->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_kkt_factor_solve(hipSetDevice)");
  hipStream_t s = (hipStream_t)stream;
  const Regions r = regions(p, d_work);
  const bool fused_rhs = p->staged || p->chain_kernels;
  // The dynamics Jacobians stay where the model callback left them: the sweep reads them in place, at whatever
  // 8-byte aligned offsets and strides the arena puts them (odd m: odd ones; launch_qw16_split)
  const long ab_off = (long)p->ck.node_len + p->ck.n * p->ck.n + p->ck.n * p->ck.m + p->ck.m * p->ck.m;
  const long ab_stage = (long)p->ck.node_len + p->ck.edge_len;
  const bool split = p->chain_split && fused_rhs && p->E > 0 && ((uintptr_t)d_model & 7) == 0;
  const bool sym = split && p->chain_sym != nullptr;
  hipError_t e = launch_condense(p, r, d_model, d_w, d_r1, d_r2, d_r3, fused_rhs ? d_b : nullptr, s, split, sym);
  if (e == hipSuccess && !fused_rhs)
    e = launch_rhs(p, r, d_model, d_b, nullptr, s);
  if (e != hipSuccess)
    return report(e, "sip_kkt_factor_solve(condense)");
  const int rc = split ? sip_lqr_factor_solve_split(sym ? p->chain_sym : p->chain, r.in0, d_model + ab_off,
                                                    p->ck.model_len, ab_stage, r.in1, r.out, r.gain, d_status, r.lqr, s)
                       : sip_lqr_factor_solve(p->chain, r.in0, r.in1, r.out, r.gain, d_status, r.lqr, s);
  if (rc != SIP_LQR_OK)
    return rc;
  e = launch_merge(p, r, d_status, s);
  if (e == hipSuccess)
    e = launch_recover(p, r, d_model, d_b, d_sol, d_status, s);
  return report(e, "sip_kkt_factor_solve(recover)");
}

int sip_kkt_add_Kx_to_y(const sip_kkt_plan *p, const double *d_model, const double *d_w, const double *d_r1,
                        const double *d_r2, const double *d_r3, const double *d_x, double *d_y, void *stream) {
  if (p == nullptr || p->input_status != SIP_KKT_SUCCESS)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const long kkt = (long)p->x_dim + p->y_dim + p->z_dim;
  if (kkt == 0)
    return SIP_LQR_OK;
  if ((!d_model && p->model_len > 0) || (!d_w && p->z_dim > 0) || (!d_r1 && p->x_dim > 0) ||
      (!d_r2 && p->y_dim > 0) || (!d_r3 && p->z_dim > 0) || !d_x || !d_y)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  return apply_blocks(p, d_model, nullptr, d_w, d_r1, d_r2, d_r3, kkt_vector_io(p, 0, d_x, d_y), (hipStream_t)stream,
                      "sip_kkt_add_Kx_to_y");
}

// The five block operators (helpers.hpp:20-24): one bit of ApplyPart each, vectors by space.
#define SIP_KKT_BLOCK_OP(NAME, PART, XS, YS)                                                                 \
  int sip_kkt_add_##NAME##_to_y(const sip_kkt_plan *p, const double *d_model, const double *d_x, double *d_y,  \
                                void *stream) {                                                              \
    return block_op(p, d_model, nullptr, sipamd::kkt::PART, XS, YS, d_x, d_y, (hipStream_t)stream,            \
                    "sip_kkt_add_" #NAME "_to_y");                                                           \
  }                                                                                                          \
  int sip_kkt_add_##NAME##_to_y_theta(const sip_kkt_plan *p, const double *d_model, const double *d_theta,     \
                                      const double *d_x, double *d_y, void *stream) {                        \
    if (p == nullptr || p->theta_dim < 1 || d_theta == nullptr)                                             \
      return SIP_LQR_ERR_INVALID_ARGUMENT;                                                                   \
    return block_op(p, d_model, d_theta, sipamd::kkt::PART, XS, YS, d_x, d_y, (hipStream_t)stream,            \
                    "sip_kkt_add_" #NAME "_to_y_theta");                                                     \
  }
SIP_KKT_BLOCK_OP(Hx, AP_H, 0, 0)
SIP_KKT_BLOCK_OP(Cx, AP_C, 0, 1)
SIP_KKT_BLOCK_OP(CTx, AP_CT, 1, 0)
SIP_KKT_BLOCK_OP(Gx, AP_G, 0, 2)
SIP_KKT_BLOCK_OP(GTx, AP_GT, 2, 0)
#undef SIP_KKT_BLOCK_OP

} // extern "C"

// ---------------------------------------------------------------------------
// theta (global variables): Schur complement around the stagewise solve
// ---------------------------------------------------------------------------
namespace {

struct ThetaRegions {
  double *J, *KJ, *S, *rhs_sw, *sol_sw, *vecs_cols, *lsol_cols; // the last two: chain plans only
  void *cws; // column workspace of sip_lqr_solve_multi (chain plans)
};
size_t theta_j_scalars(const sip_kkt_plan *p) { // per problem
  const size_t skkt = (size_t)p->x_dim + p->y_dim + p->z_dim, th = (size_t)p->theta_dim;
  return p->chain_theta ? (size_t)p->N * (th * th + th) : skkt * th;
}
ThetaRegions theta_regions(const sip_kkt_plan *p, void *theta_work) {
  const size_t skkt = (size_t)p->x_dim + p->y_dim + p->z_dim, B = (size_t)p->batch;
  char *w = (char *)theta_work;
  ThetaRegions r;
  size_t cur = 0;
  // (the fused chain passes keep stage partials here instead of J_theta: S_part (N p^2) then d_part (N p) per problem)
  r.J = (double *)(w + cur), cur = align256(cur + sizeof(double) * B * theta_j_scalars(p));
  r.KJ = (double *)(w + cur), cur = align256(cur + sizeof(double) * B * skkt * p->theta_dim);
  r.S = (double *)(w + cur), cur = align256(cur + sizeof(double) * B * p->theta_dim * p->theta_dim);
  r.rhs_sw = (double *)(w + cur), cur = align256(cur + sizeof(double) * B * skkt);
  r.sol_sw = (double *)(w + cur), cur = align256(cur + sizeof(double) * B * skkt);
  const size_t cols = p->chain_kernels ? sizeof(double) * B * (size_t)p->in1_len * p->theta_dim : 0;
  r.vecs_cols = (double *)(w + cur), cur = align256(cur + cols);
  r.lsol_cols = (double *)(w + cur), cur = align256(cur + cols);
  r.cws = w + cur;
  return r;
}

} // namespace

extern "C" {

int sip_kkt_plan_set_theta(sip_kkt_plan *p, int theta_dim) {
  if (p == nullptr || theta_dim < 1 || p->theta_dim != 0 || p->input_status != SIP_KKT_SUCCESS)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  using namespace sipamd::kkt;
  const int E = p->E, N = p->N, th = theta_dim;
  for (auto &v : p->toff)
    v.assign(N, 0);
  long at = 0;
  for (int i = 0; i < N; ++i) {
    p->toff[TH_N_X][i] = at, at += (long)p->sd[i] * th;
    p->toff[TH_N_C][i] = at, at += (long)p->ncd[i] * th;
    p->toff[TH_N_G][i] = at, at += (long)p->ngd[i] * th;
    p->toff[TH_N_TT][i] = at, at += (long)th * th;
    if (i < E) {
      const int e = i;
      p->toff[TH_E_X][e] = at, at += (long)p->sd[p->parents[e]] * th;
      p->toff[TH_E_U][e] = at, at += (long)p->cd[e] * th;
      p->toff[TH_E_DYN][e] = at, at += (long)p->sd[p->children[e]] * th;
      p->toff[TH_E_C][e] = at, at += (long)p->ecd[e] * th;
      p->toff[TH_E_G][e] = at, at += (long)p->egd[e] * th;
      p->toff[TH_E_TT][e] = at, at += (long)th * th;
    }
  }
  std::vector<long> longs;
  size_t where[TH_NUM_BLOCKS];
  for (int b = 0; b < TH_NUM_BLOCKS; ++b) {
    where[b] = longs.size();
    longs.insert(longs.end(), p->toff[b].begin(), p->toff[b].end());
  }
  sipamd::DeviceGuard on_device(p->device);
  hipError_t he = on_device.err;
  if (he == hipSuccess)
    he = hipMalloc(&p->d_theta_longs, longs.size() * sizeof(long));
  if (he == hipSuccess)
    he = hipMemcpy(p->d_theta_longs, longs.data(), longs.size() * sizeof(long), hipMemcpyHostToDevice);
  if (he != hipSuccess)
    return report(he, "sip_kkt_plan_set_theta");
  p->theta_len = at, p->theta_dim = th;
  p->theta_meta.p = th, p->theta_meta.theta_len = at;
  if (p->chain_kernels && E >= 1) {
    const sipamd::kkt::ChainKkt &ck = p->ck;
    sipamd::kkt::ChainTheta &ct = p->ct;
    ct.p = th, ct.theta_len = at;
    ct.node_len = (ck.n + ck.cn + ck.gn + th) * th;
    ct.edge_len = (2 * ck.n + ck.m + ck.ce + ck.ge + th) * th;
    ct.lds_item = even(std::max(ct.node_len + ct.edge_len, (ck.n + ck.cT + ck.gT + th) * th));
    bool as_assumed = true; // the blocks of stage i are one item at i (node_len + edge_len), in ThetaBlock order
    for (int i = 0; i < N && as_assumed; ++i) {
      const long base = (long)i * (ct.node_len + ct.edge_len);
      const int c = i == E ? ck.cT : ck.cn, g = i == E ? ck.gT : ck.gn;
      as_assumed = p->toff[TH_N_X][i] == base && p->toff[TH_N_C][i] == base + (long)ck.n * th &&
                   p->toff[TH_N_G][i] == base + (long)(ck.n + c) * th &&
                   p->toff[TH_N_TT][i] == base + (long)(ck.n + c + g) * th;
      if (as_assumed && i < E)
        as_assumed = p->toff[TH_E_X][i] == base + ct.node_len && p->toff[TH_E_U][i] == base + ct.node_len + (long)ck.n * th &&
                     p->toff[TH_E_DYN][i] == base + ct.node_len + (long)(ck.n + ck.m) * th &&
                     p->toff[TH_E_C][i] == base + ct.node_len + (long)(2 * ck.n + ck.m) * th &&
                     p->toff[TH_E_G][i] == base + ct.node_len + (long)(2 * ck.n + ck.m + ck.ce) * th &&
                     p->toff[TH_E_TT][i] == base + ct.node_len + (long)(2 * ck.n + ck.m + ck.ce + ck.ge) * th &&
                     p->parents[i] == i && p->children[i] == i + 1;
    }
    const size_t per = (size_t)ck.n + ck.m, R = (size_t)ck.lds_rows, P = (size_t)th;
    p->lds_theta_rhs = sizeof(double) * ((size_t)ck.lds_tail + ct.lds_item + R + P * R);
    p->lds_theta_recover = sizeof(double) * ((size_t)ck.lds_tail + ct.lds_item + P * per + P * ck.n + P * R + R);
    p->lds_theta_dot = sizeof(double) * ((size_t)ct.lds_item + per + ck.n + R);
    const char *fe = std::getenv("SIP_KKT_THETA_FUSED");
    p->chain_theta = as_assumed && std::max(p->lds_theta_rhs, p->lds_theta_recover) <= 64 * 1024 &&
                     !(fe != nullptr && fe[0] == '0');
    if (p->chain_theta)
      p->name += " + fused theta passes";
  }
  for (int b = 0; b < TH_NUM_BLOCKS; ++b)
    p->theta_meta.to[b] = (const long *)p->d_theta_longs + where[b];
  return SIP_LQR_OK;
}

size_t sip_kkt_theta_len(const sip_kkt_plan *p) { return (p && p->theta_dim > 0) ? (size_t)p->theta_len : 0; }

size_t sip_kkt_theta_offset(const sip_kkt_plan *p, int block, int index) {
  if (p == nullptr || p->theta_dim < 1 || block < 0 || block >= SIP_KKT_TH_NUM_BLOCKS || index < 0 ||
      index >= (block <= SIP_KKT_TH_NODE_D2L_DTHETA2 ? p->N : p->E))
    return (size_t)-1;
  return (size_t)p->toff[block][index];
}

size_t sip_kkt_theta_work_bytes(const sip_kkt_plan *p) {
  if (p == nullptr || p->theta_dim < 1)
    return 0;
  const size_t skkt = (size_t)p->x_dim + p->y_dim + p->z_dim, B = (size_t)p->batch, th = (size_t)p->theta_dim;
  const size_t cols = p->chain_kernels ? align256(sizeof(double) * B * (size_t)p->in1_len * th) : 0;
  const size_t cws = p->chain_kernels ? align256(sip_lqr_solve_multi_workspace_bytes(p->chain, p->theta_dim)) : 0;
  return align256(sizeof(double) * B * theta_j_scalars(p)) + align256(sizeof(double) * B * skkt * th) +
         align256(sizeof(double) * B * th * th) + 2 * align256(sizeof(double) * B * skkt) + 2 * cols + cws;
}

int sip_kkt_factor_theta(const sip_kkt_plan *p, const double *d_model, const double *d_theta, const double *d_w,
                         const double *d_r1, const double *d_r2, const double *d_r3, void *d_work,
                         void *d_theta_work, int32_t *d_status, void *stream) {
  if (p == nullptr || p->theta_dim < 1 || !d_theta || !d_theta_work || !d_r1)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  // r1 of problem q starts at q * (x_dim + p); the condensation kernels index r1 with the
  // stagewise stride x_dim, so they get a compacted copy (theta entries dropped).
  sipamd::DeviceGuard on_device(p->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_kkt_factor_theta(hipSetDevice)");
  hipStream_t s = (hipStream_t)stream;
  const ThetaRegions t = theta_regions(p, d_theta_work);
  const int sx = p->x_dim, th = p->theta_dim;
  const long skkt = (long)sx + p->y_dim + p->z_dim;
  // compact r1 (drop the theta entries) into rhs_sw's space (free until solve)
  // (a kernel, not a memcpy node: the entry points stay graph-capturable, stream_fill.hpp)
  if (sx > 0)
    hipLaunchKernelGGL(sipamd::kkt::theta_strip_kernel, dim3((unsigned)((p->batch * (long)sx + 255) / 256)), dim3(256), 0,
                       s, d_r1, t.rhs_sw, sx, th, (long)sx, (long)p->batch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return report(e, "sip_kkt_factor_theta(r1)");
  int rc = sip_kkt_factor(p, d_model, d_w, t.rhs_sw, d_r2, d_r3, d_work, d_status, stream);
  if (rc != SIP_LQR_OK)
    return rc;
  if (p->chain_theta) {
    // the fused passes of kkt_theta_chain_kernels.hpp: right-hand sides of all columns straight from the theta
    // arena, one multi-rhs Riccati solve, multipliers of all columns + the stage shares of the Schur complement,
    // their sum and its LLT
    const Regions r = regions(p, d_work);
    const long colJ = (long)p->batch * skkt, colV = (long)p->batch * p->in1_len;
    double *s_part = t.J;
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::theta_rhs_chain_kernel<decltype(fn)::value, decltype(fm)::value>),
                         dim3(node_grid(p)), dim3(sipamd::kkt::TPB), p->lds_theta_rhs, s, p->ck, p->ct, d_model, d_theta,
                         (const double *)r.inv, t.vecs_cols, colV, (const int32_t *)d_status, (long)p->batch);
    });
    if ((e = hipGetLastError()) != hipSuccess)
      return report(e, "sip_kkt_factor_theta(rhs)");
    rc = sip_lqr_solve_multi(p->chain, r.in0, t.vecs_cols, t.lsol_cols, th, r.gain, r.lqr, t.cws, s);
    if (rc != SIP_LQR_OK)
      return rc;
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::theta_recover_chain_kernel<decltype(fn)::value, decltype(fm)::value>),
                         dim3(node_grid(p)), dim3(sipamd::kkt::TPB), p->lds_theta_recover, s, p->ck, p->ct, d_model,
                         d_theta, (const double *)r.inv, (const double *)t.lsol_cols, colV, t.KJ, colJ, s_part,
                         (const int32_t *)d_status, (long)p->batch);
    });
    if ((e = hipGetLastError()) != hipSuccess)
      return report(e, "sip_kkt_factor_theta(recover)");
    hipLaunchKernelGGL(sipamd::kkt::theta_schur_reduce_kernel, dim3((unsigned)p->batch), dim3(sipamd::kkt::TPB),
                       sizeof(double) * (size_t)th * th, s, p->N, th, sx, d_r1, (const double *)s_part, t.S, d_status,
                       (long)p->batch, (int)SIP_KKT_THETA_SCHUR_FAILURE);
    return report(hipGetLastError(), "sip_kkt_factor_theta(schur)");
  }
  hipLaunchKernelGGL(sipamd::kkt::theta_jacobian_kernel, dim3(node_grid(p)), dim3(sipamd::kkt::TPB), 0, s, p->meta,
                     p->theta_meta, d_theta, t.J, (long)p->batch);
  if ((e = hipGetLastError()) != hipSuccess)
    return report(e, "sip_kkt_factor_theta(jacobian)");
  if (p->chain_kernels) {
    // K^-1 J_theta (helpers.cpp:387): the right-hand sides of all columns from one staging of the
    // Jacobians, one Riccati solve per column, the multipliers of all columns from one staging
    const Regions r = regions(p, d_work);
    const long colJ = (long)p->batch * skkt, colV = (long)p->batch * p->in1_len;
    // (LDS: one block of weighted rows per column; as many columns per launch as 64 KiB hold)
    const size_t rhs_col_lds = sizeof(double) * (size_t)p->ck.lds_rows;
    const size_t rhs_lds = sizeof(double) * ((size_t)p->ck.lds_tail + 2 * (size_t)p->ck.lds_rows); // one column
    const int rhs_cols = 1 + (int)std::min<size_t>((size_t)(th - 1), (64 * 1024 - rhs_lds) / std::max<size_t>(rhs_col_lds, 1));
    for (int c0 = 0; c0 < th; c0 += rhs_cols) {
      const int nc = std::min(rhs_cols, th - c0);
      family_dispatch(p->family, [&](auto fn, auto fm) {
        hipLaunchKernelGGL((sipamd::kkt::condense_chain_kernel<true, false, decltype(fn)::value, decltype(fm)::value>),
                           dim3(node_grid(p)), dim3(sipamd::kkt::TPB),
                           rhs_lds + rhs_col_lds * (size_t)(nc - 1), s, p->ck, d_model,
                           (const double *)nullptr, r.inv, r.in0, (const double *)t.J + (size_t)c0 * colJ,
                           t.vecs_cols + (size_t)c0 * colV, (long)p->batch, (const int32_t *)d_status, nc, colJ, colV);
      });
    }
    if ((e = hipGetLastError()) != hipSuccess)
      return report(e, "sip_kkt_factor_theta(rhs)");
    // all columns through one backward / forward sweep where the shape has the multi-rhs kernel
    // (chain_mrhs.hpp: the GEMM form of helpers.cpp:521-665), column by column otherwise
    rc = sip_lqr_solve_multi(p->chain, r.in0, t.vecs_cols, t.lsol_cols, th, r.gain, r.lqr, t.cws, s);
    if (rc != SIP_LQR_OK)
      return rc;
    // (LDS: x_i | u_i of every column of a launch)
    const size_t rec_col_lds = sizeof(double) * (size_t)(p->ck.n + p->ck.m);
    const int rec_cols = 1 + (int)std::min<size_t>((size_t)(th - 1), (64 * 1024 - p->lds_chain_recover) / rec_col_lds);
    for (int c0 = 0; c0 < th; c0 += rec_cols) {
      const int nc = std::min(rec_cols, th - c0);
      if (nc > 1)
        family_dispatch(p->family, [&](auto fn, auto fm) {
          hipLaunchKernelGGL((sipamd::kkt::recover_chain_kernel<true, decltype(fn)::value, decltype(fm)::value>),
                             dim3(node_grid(p)), dim3(sipamd::kkt::TPB),
                             p->lds_chain_recover + rec_col_lds * (size_t)(nc - 1), s, p->ck, d_model,
                             (const double *)t.J + (size_t)c0 * colJ, r.inv,
                             (const double *)t.lsol_cols + (size_t)c0 * colV, t.KJ + (size_t)c0 * colJ,
                             (const int32_t *)d_status, (long)p->batch, nc, colJ, colV, colJ);
        });
      else
        family_dispatch(p->family, [&](auto fn, auto fm) {
          hipLaunchKernelGGL((sipamd::kkt::recover_chain_kernel<false, decltype(fn)::value, decltype(fm)::value>),
                             dim3(node_grid(p)), dim3(sipamd::kkt::TPB), p->lds_chain_recover, s, p->ck, d_model,
                             (const double *)t.J + (size_t)c0 * colJ, r.inv,
                             (const double *)t.lsol_cols + (size_t)c0 * colV, t.KJ + (size_t)c0 * colJ,
                             (const int32_t *)d_status, (long)p->batch, 1, colJ, colV, colJ);
        });
    }
    if ((e = hipGetLastError()) != hipSuccess)
      return report(e, "sip_kkt_factor_theta(recover)");
  } else {
    for (int col = 0; col < th; ++col) { // K^-1 J_theta, one column per launch (helpers.cpp:387)
      rc = sip_kkt_solve(p, d_model, t.J + (size_t)col * p->batch * skkt, t.KJ + (size_t)col * p->batch * skkt,
                         d_work, d_status, stream);
      if (rc != SIP_LQR_OK)
        return rc;
    }
  }
  if (th <= 8)
    hipLaunchKernelGGL(sipamd::kkt::theta_schur_small_kernel<8>, dim3((unsigned)p->batch), dim3(sipamd::kkt::TPB),
                       0, s, p->meta, p->theta_meta, d_theta, d_r1, t.J, t.KJ, t.S, d_status, (long)p->batch,
                       (int)SIP_KKT_THETA_SCHUR_FAILURE);
  else
    hipLaunchKernelGGL(sipamd::kkt::theta_schur_kernel, dim3((unsigned)p->batch), dim3(sipamd::kkt::TPB),
                       sizeof(double) * (size_t)th * th, s, p->meta, p->theta_meta, d_theta, d_r1, t.J, t.KJ, t.S,
                       d_status, (long)p->batch, (int)SIP_KKT_THETA_SCHUR_FAILURE);
  return report(hipGetLastError(), "sip_kkt_factor_theta(schur)");
}

int sip_kkt_solve_theta(const sip_kkt_plan *p, const double *d_model, const double *d_theta, const double *d_b,
                        double *d_sol, void *d_work, void *d_theta_work, const int32_t *d_status, void *stream) {
  if (p == nullptr || p->theta_dim < 1 || !d_theta || !d_theta_work || !d_b || !d_sol || !d_status)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sipamd::DeviceGuard on_device(p->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return report(on_device.err, "sip_kkt_solve_theta(hipSetDevice)");
  hipStream_t s = (hipStream_t)stream;
  const ThetaRegions t = theta_regions(p, d_theta_work);
  const int sx = p->x_dim, th = p->theta_dim;
  const long skkt = (long)sx + p->y_dim + p->z_dim;
  hipLaunchKernelGGL(sipamd::kkt::theta_strip_kernel, dim3((unsigned)((p->batch * skkt + 255) / 256)), dim3(256), 0, s,
                     d_b, t.rhs_sw, sx, th, skkt, (long)p->batch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return report(e, "sip_kkt_solve_theta(strip)");
  const int rc = sip_kkt_solve(p, d_model, t.rhs_sw, t.sol_sw, d_work, d_status, stream);
  if (rc != SIP_LQR_OK)
    return rc;
  if (p->chain_theta) {
    double *d_part = t.J + (size_t)p->batch * p->N * th * th; // behind the Schur partials (kept: solve after solve)
    family_dispatch(p->family, [&](auto fn, auto fm) {
      hipLaunchKernelGGL((sipamd::kkt::theta_dot_chain_kernel<decltype(fn)::value, decltype(fm)::value>),
                         dim3(node_grid(p)), dim3(sipamd::kkt::TPB), p->lds_theta_dot, s, p->ck, p->ct, d_theta,
                         (const double *)t.sol_sw, d_part, d_status, (long)p->batch);
    });
    hipLaunchKernelGGL(sipamd::kkt::theta_finish_parts_kernel, dim3((unsigned)p->batch), dim3(sipamd::kkt::TPB),
                       sizeof(double) * (size_t)th, s, p->N, th, sx, skkt, d_b, (const double *)d_part,
                       (const double *)t.KJ, (const double *)t.S, (const double *)t.sol_sw, d_sol, d_status,
                       (long)p->batch);
    return report(hipGetLastError(), "sip_kkt_solve_theta(finish)");
  }
  hipLaunchKernelGGL(sipamd::kkt::theta_finish_kernel, dim3((unsigned)p->batch), dim3(sipamd::kkt::TPB),
                     sizeof(double) * (size_t)th, s, p->meta, p->theta_meta, d_b, t.J, t.KJ, t.S, t.sol_sw, d_sol,
                     d_status, (long)p->batch);
  return report(hipGetLastError(), "sip_kkt_solve_theta(finish)");
}

int sip_kkt_add_Kx_to_y_theta(const sip_kkt_plan *p, const double *d_model, const double *d_theta,
                              const double *d_w, const double *d_r1, const double *d_r2, const double *d_r3,
                              const double *d_x, double *d_y, void *stream) {
  if (p == nullptr || p->theta_dim < 1 || !d_theta || !d_r1 || !d_x || !d_y)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  // the stagewise operator on [x | theta | y | z] vectors, then the theta terms
  return apply_blocks(p, d_model, d_theta, d_w, d_r1, d_r2, d_r3, kkt_vector_io(p, p->theta_dim, d_x, d_y),
                      (hipStream_t)stream, "sip_kkt_add_Kx_to_y_theta");
}

#ifdef SIP_KKT_STAMPS
// diagnostic build: read (and clear) the per-segment cycle sums of condense_chain_pipe_kernel
void sip_kkt_debug_segments(unsigned long long *out16) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(sipamd::kkt::g_kkt_seg), 16 * sizeof(unsigned long long));
  unsigned long long zero[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(sipamd::kkt::g_kkt_seg), zero, sizeof(zero));
}
#endif

} // extern "C"
