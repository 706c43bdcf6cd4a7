// sip_lqr_tree.hip -- C ABI of the general tree / variable-dimension path
// (include/sip_lqr_amd.h, second half) over csrc/tree_generic.hpp.
#include "../../include/sip_lqr_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "generic_plan.hpp"
#include "tree_qw16_launch.hpp"

struct sip_lqr_tree_plan {
  int64_t batch = 0;
  int device = 0;
  int topology_status = SIP_LQR_INVALID_TOPOLOGY;
  sipamd::GenericPlan g;
  // fused factor + solve on the padded size class (tree_qw16.hpp); nullptr: general engine only
  const sipamd::TreeClass *fused = nullptr;
  sipamd::TreeSchedule sched{};
  void *d_steps = nullptr;
  size_t at_spill = 0, scratch_bytes = 0; // scratch: padded gains at 0, then the spill
  ~sip_lqr_tree_plan() {
    if (d_steps != nullptr)
      (void)hipFree(d_steps);
  }
};

extern "C" {

// Host integer work, exactly the reference's traversal compilation
// (lqr.cpp:563-631): counting sort of the edges by parent, then an iterative
// DFS that pushes children in reverse so the lowest edge index is visited
// first; postorder is the reversed preorder.
int sip_lqr_compile_topology(int num_edges, int root, const int *edge_parents,
                             const int *edge_children, int *child_offsets,
                             int *child_edges, int *edge_parents_out,
                             int *edge_children_out, int *preorder,
                             int *postorder, int *node_marks) {
  const int E = num_edges, N = num_edges + 1;
  if (E < 0 || edge_parents == nullptr || edge_children == nullptr)
    return SIP_LQR_INVALID_TOPOLOGY;
  if (root < 0 || root >= N)
    return SIP_LQR_INVALID_TOPOLOGY;
  std::fill(child_offsets, child_offsets + N + 1, 0);
  for (int e = 0; e < E; ++e) {
    const int p = edge_parents[e], c = edge_children[e];
    if (p < 0 || p >= N || c < 0 || c >= N || p == c)
      return SIP_LQR_INVALID_TOPOLOGY;
    edge_parents_out[e] = p;
    edge_children_out[e] = c;
    ++child_offsets[p + 1];
  }
  for (int node = 0; node < N; ++node)
    child_offsets[node + 1] += child_offsets[node];
  std::vector<int> cursor(child_offsets, child_offsets + N);
  for (int e = 0; e < E; ++e)
    child_edges[cursor[edge_parents_out[e]]++] = e;

  std::vector<int> stack;
  stack.reserve(N);
  stack.push_back(root);
  std::fill(node_marks, node_marks + N, 0);
  int visited = 0;
  while (!stack.empty()) {
    const int node = stack.back();
    stack.pop_back();
    if (visited >= N || node_marks[node] != 0)
      return SIP_LQR_INVALID_TOPOLOGY; // cycle or two parents
    node_marks[node] = 1;
    preorder[visited++] = node;
    for (int ci = child_offsets[node + 1] - 1; ci >= child_offsets[node]; --ci)
      stack.push_back(edge_children_out[child_edges[ci]]);
  }
  if (visited != N)
    return SIP_LQR_INVALID_TOPOLOGY; // disconnected
  for (int order = 0; order < N; ++order)
    postorder[order] = preorder[N - 1 - order];
  return SIP_LQR_SUCCESS;
}

int sip_lqr_tree_plan_create(int64_t batch, int num_edges, int root,
                             const int *edge_parents, const int *edge_children,
                             const int *state_dims, const int *control_dims,
                             int device, sip_lqr_tree_plan **out) {
  if (out == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (batch < 1 || num_edges < 0 || state_dims == nullptr ||
      (num_edges > 0 && control_dims == nullptr))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const int E = num_edges, N = E + 1;
  for (int i = 0; i < N; ++i)
    if (state_dims[i] < 0)
      return SIP_LQR_ERR_INVALID_ARGUMENT;
  for (int e = 0; e < E; ++e)
    if (control_dims[e] < 0)
      return SIP_LQR_ERR_INVALID_ARGUMENT;
  sip_lqr_tree_plan *p = new (std::nothrow) sip_lqr_tree_plan;
  if (p == nullptr)
    return SIP_LQR_ERR_ALLOC;
  p->batch = batch;
  p->device = device;
  sipamd::GenericPlan &g = p->g;
  g.set_shape(E, root, state_dims, control_dims);
  g.parents.assign(E, 0), g.children.assign(E, 0);
  g.child_offsets.assign(N + 1, 0), g.child_edges.assign(E, 0);
  g.preorder.assign(N, 0), g.postorder.assign(N, 0);
  std::vector<int> marks(N, 0);
  p->topology_status = sip_lqr_compile_topology(
      E, root, edge_parents, edge_children, g.child_offsets.data(), g.child_edges.data(),
      g.parents.data(), g.children.data(), g.preorder.data(), g.postorder.data(), marks.data());
  *out = p;
  if (p->topology_status != SIP_LQR_SUCCESS)
    return SIP_LQR_OK; // latched, reported by factor (lqr.cpp:646-648)
  g.layout_tree_native();
  const hipError_t e = g.upload(device);
  if (e != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_tree_plan_create: HIP error: %s\n", hipGetErrorString(e));
    delete p;
    *out = nullptr;
    return SIP_LQR_ERR_HIP;
  }
  // fused path: a padded size class of the broadcast-FMA tree kernel, when the tree fits one
  const char *variant = std::getenv("SIP_LQR_TREE");
  const bool general_only = variant != nullptr && (std::strcmp(variant, "general") == 0 || std::strcmp(variant, "global") == 0);
  // (its clamped loads need one readable scalar in the input and the output arena)
  p->fused = (general_only || g.in0_len < 1 || g.out_len < 1)
                 ? nullptr
                 : sipamd::find_tree_class(std::max(1, g.max_n), std::max(1, g.max_m));
  if (p->fused != nullptr) {
    // the flattened traversal (TreeStep records, tree_qw16.hpp), from the host copies of the tables
    std::vector<sipamd::TreeStep> steps;
    auto node_fields = [&](sipamd::TreeStep &st, int j) {
      st.node = j, st.n = g.state_dims[j];
      st.oQ = g.oQ[j], st.oq = g.oq[j], st.oc = g.oc[j], st.od = g.od[j];
      st.oV = g.oV[j], st.oF = g.oF[j], st.osd = g.osd[j], st.osdi = g.osdi[j], st.ov = g.ov[j];
    };
    auto edge_fields = [&](sipamd::TreeStep &st, int e) {
      const int ch = g.children[e];
      st.edge = e, st.child = ch, st.nc = g.state_dims[ch], st.m = g.control_dims[e];
      st.oA = g.oA[e], st.oB = g.oB[e], st.oM = g.oM[e], st.oR = g.oR[e], st.orr = g.orr[e], st.odc = g.od[ch];
      st.oK = g.oK[e], st.ok = g.ok[e], st.ou = g.ou[e], st.oxc = g.ox[ch], st.oyc = g.oy[ch];
      st.oW = g.oW[e], st.oG = g.oG[e];
      st.oxp = g.ox[g.parents[e]];
    };
    int finished = -1; // node finished by the previous backward step
    for (int idx = 0; idx < N; ++idx) {
      const int j = g.postorder[idx];
      const int lo = g.child_offsets[j], hi = g.child_offsets[j + 1];
      for (int ci = lo; ci < hi; ++ci) {
        sipamd::TreeStep st{};
        st.kind = 0;
        node_fields(st, j);
        edge_fields(st, g.child_edges[ci]);
        st.flags = (ci == lo ? sipamd::TS_LOAD_V : 0) | (st.child == finished ? sipamd::TS_CHILD_LIVE : 0);
        steps.push_back(st);
        finished = -1;
      }
      sipamd::TreeStep st{};
      st.kind = 1;
      node_fields(st, j);
      st.flags = lo == hi ? sipamd::TS_LOAD_V : 0;
      steps.push_back(st);
      finished = j;
    }
    const size_t n_backward = steps.size();
    int produced = root; // node whose x the previous forward step produced (the root's comes first)
    for (int idx = 0; idx < N; ++idx) {
      const int j = g.preorder[idx];
      for (int ci = g.child_offsets[j]; ci < g.child_offsets[j + 1]; ++ci) {
        sipamd::TreeStep st{};
        node_fields(st, j);
        edge_fields(st, g.child_edges[ci]);
        st.flags = j == produced ? sipamd::TS_CHILD_LIVE : 0;
        steps.push_back(st);
        produced = st.child;
      }
    }
    {
      sipamd::DeviceGuard on_device(device);
      hipError_t he = on_device.err;
      if (he == hipSuccess)
        he = hipMalloc(&p->d_steps, std::max<size_t>(1, steps.size()) * sizeof(sipamd::TreeStep));
      if (he == hipSuccess)
        he = hipMemcpy(p->d_steps, steps.data(), steps.size() * sizeof(sipamd::TreeStep), hipMemcpyHostToDevice);
      if (he != hipSuccess) {
        std::fprintf(stderr, "sip_lqr_tree_plan_create: HIP error: %s\n", hipGetErrorString(he));
        delete p;
        *out = nullptr;
        return SIP_LQR_ERR_HIP;
      }
    }
    sipamd::TreeSchedule &t = p->sched;
    t.backward = (const sipamd::TreeStep *)p->d_steps, t.forward = t.backward + n_backward;
    t.n_backward = (int)n_backward, t.n_forward = (int)(steps.size() - n_backward);
    t.Nn = N, t.E = E, t.root = root, t.root_n = g.state_dims[root];
    t.root_od = g.od[root], t.root_ox = g.ox[root], t.root_oy = g.oy[root];
    t.in_len = g.in0_len, t.out_len = g.out_len, t.ws_len = g.ws_len;
    const size_t PN = p->fused->n, PM = p->fused->m, B = (size_t)batch * sizeof(double);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t cur = up(B * ((size_t)E * (PM * PN + PM))); // padded gains at 0
    p->at_spill = cur, cur = up(cur + B * ((size_t)N * (PN * PN + 4 * PN)));
    p->scratch_bytes = cur;
  }
  return SIP_LQR_OK;
}

void sip_lqr_tree_plan_destroy(sip_lqr_tree_plan *plan) { delete plan; }

int sip_lqr_tree_topology_status(const sip_lqr_tree_plan *plan) {
  return plan ? plan->topology_status : SIP_LQR_INVALID_TOPOLOGY;
}

const int *sip_lqr_tree_topology_array(const sip_lqr_tree_plan *plan, int which) {
  if (plan == nullptr)
    return nullptr;
  switch (which) {
  case 0: return plan->g.child_offsets.data();
  case 1: return plan->g.child_edges.data();
  case 2: return plan->g.preorder.data();
  case 3: return plan->g.postorder.data();
  default: return nullptr;
  }
}

size_t sip_lqr_tree_input_len(const sip_lqr_tree_plan *p) { return p ? (size_t)p->g.in0_len : 0; }
size_t sip_lqr_tree_work_len(const sip_lqr_tree_plan *p) { return p ? (size_t)p->g.ws_len : 0; }
size_t sip_lqr_tree_output_len(const sip_lqr_tree_plan *p) { return p ? (size_t)p->g.out_len : 0; }

size_t sip_lqr_tree_offset(const sip_lqr_tree_plan *p, int arena, int kind, int index) {
  if (p == nullptr || p->topology_status != SIP_LQR_SUCCESS || arena < 0 || arena > 2 ||
      kind < 0 || kind > 1 || index < 0 || index >= (kind == 0 ? p->g.N : p->g.E))
    return (size_t)-1;
  // first block of each (arena, kind) group: Q.. / A.. ; W.. / V.. ; x.. / u
  const std::vector<long> *tab[3][2] = {{&p->g.oQ, &p->g.oA}, {&p->g.oV, &p->g.oW}, {&p->g.ox, &p->g.ou}};
  return (size_t)(*tab[arena][kind])[index];
}

int sip_lqr_tree_factor(const sip_lqr_tree_plan *plan, const double *d_input,
                        double *d_work, int32_t *d_status, void *stream) {
  if (plan == nullptr || d_status == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  sipamd::DeviceGuard on_device(plan->device); // launch on the plan's device, whatever the caller's current one
  if (on_device.err != hipSuccess)
    return SIP_LQR_ERR_HIP;
  if (plan->topology_status != SIP_LQR_SUCCESS) {
    // every instance reports the latched traversal status (lqr.cpp:646-648)
    std::vector<int32_t> st((size_t)plan->batch, plan->topology_status);
    if (hipMemcpyAsync(d_status, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return SIP_LQR_ERR_HIP;
    return SIP_LQR_OK;
  }
  if ((d_input == nullptr && plan->g.in0_len > 0) || d_work == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  const hipError_t e = plan->g.launch_factor<double>((long)plan->batch, d_input, d_work, d_work, d_status, s);
  if (e != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_tree_factor: %s\n", hipGetErrorString(e));
    return SIP_LQR_ERR_HIP;
  }
  return SIP_LQR_OK;
}

int sip_lqr_tree_solve(const sip_lqr_tree_plan *plan, const double *d_input,
                       double *d_work, double *d_output, const int32_t *d_status,
                       void *stream) {
  if (plan == nullptr || d_status == nullptr || d_work == nullptr ||
      (d_output == nullptr && plan->g.out_len > 0) || (d_input == nullptr && plan->g.in0_len > 0))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (plan->topology_status != SIP_LQR_SUCCESS)
    return SIP_LQR_ERR_INVALID_ARGUMENT; // solve() needs a successful factor
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return SIP_LQR_ERR_HIP;
  const hipError_t e = plan->g.launch_solve<double>((long)plan->batch, d_input, d_input, d_work, d_work,
                                                    d_output, d_status, (hipStream_t)stream);
  if (e != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_tree_solve: %s\n", hipGetErrorString(e));
    return SIP_LQR_ERR_HIP;
  }
  return SIP_LQR_OK;
}


const char *sip_lqr_tree_kernel_name(const sip_lqr_tree_plan *plan) {
  if (plan == nullptr)
    return "";
  return plan->fused != nullptr ? plan->fused->name : "tree_generic/f64";
}

size_t sip_lqr_tree_fused_scratch_bytes(const sip_lqr_tree_plan *plan) {
  return (plan != nullptr && plan->fused != nullptr) ? plan->scratch_bytes : 0;
}

namespace {
int tree_factor_solve_impl(const sip_lqr_tree_plan *plan, const double *d_input, double *d_work, double *d_output,
                           int32_t *d_status, void *d_scratch, void *stream, const bool workspace) {
  if (plan == nullptr || d_status == nullptr || (workspace && d_work == nullptr))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (plan->topology_status != SIP_LQR_SUCCESS || plan->fused == nullptr) { // general engine, two launches
    const int rc = sip_lqr_tree_factor(plan, d_input, d_work, d_status, stream);
    if (rc != SIP_LQR_OK || plan->topology_status != SIP_LQR_SUCCESS)
      return rc;
    return sip_lqr_tree_solve(plan, d_input, d_work, d_output, d_status, stream);
  }
  if ((d_input == nullptr && plan->g.in0_len > 0) || (d_output == nullptr && plan->g.out_len > 0) || d_scratch == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  sipamd::DeviceGuard on_device(plan->device);
  if (on_device.err != hipSuccess)
    return SIP_LQR_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  char *w = (char *)d_scratch;
  const auto launch = workspace ? plan->fused->launch_export : plan->fused->launch;
  const hipError_t e = launch(plan->sched, d_input, d_output, d_work, (double *)w, (double *)(w + plan->at_spill),
                              d_status, (long)plan->batch, s);
  if (e != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_tree_factor_solve: %s\n", hipGetErrorString(e));
    return SIP_LQR_ERR_HIP;
  }
  return SIP_LQR_OK;
}
} // namespace

int sip_lqr_tree_factor_solve(const sip_lqr_tree_plan *plan, const double *d_input, double *d_work, double *d_output,
                              int32_t *d_status, void *d_scratch, void *stream) {
  return tree_factor_solve_impl(plan, d_input, d_work, d_output, d_status, d_scratch, stream, false);
}

int sip_lqr_tree_factor_solve_workspace(const sip_lqr_tree_plan *plan, const double *d_input, double *d_work,
                                        double *d_output, int32_t *d_status, void *d_scratch, void *stream) {
  return tree_factor_solve_impl(plan, d_input, d_work, d_output, d_status, d_scratch, stream, true);
}

} // extern "C"
