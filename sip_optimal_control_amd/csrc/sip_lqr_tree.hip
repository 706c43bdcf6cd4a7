// sip_lqr_tree.hip -- C ABI of the general tree / variable-dimension path
// (include/sip_lqr_amd.h, second half) over csrc/tree_generic.hpp.
#include "../../include/sip_lqr_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <vector>

#include "tree_generic.hpp"

struct sip_lqr_tree_plan {
  int64_t batch = 0;
  int num_edges = 0, num_nodes = 0, root = 0, device = 0;
  int topology_status = SIP_LQR_INVALID_TOPOLOGY;
  std::vector<int> state_dims, control_dims, parents, children;
  std::vector<int> child_offsets, child_edges, preorder, postorder;
  std::vector<long> node_in, edge_in, node_ws, edge_ws, node_out, edge_out;
  long in_len = 0, ws_len = 0, out_len = 0, scratch_ws = 0;
  int max_n = 0, max_m = 0;
  void *d_ints = nullptr;  // all int tables
  void *d_longs = nullptr; // all offset tables
  sipamd::tree::Meta meta{};
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
  p->num_edges = E;
  p->num_nodes = N;
  p->root = root;
  p->device = device;
  p->state_dims.assign(state_dims, state_dims + N);
  p->control_dims.assign(control_dims, control_dims + E);
  p->parents.assign(E, 0);
  p->children.assign(E, 0);
  p->child_offsets.assign(N + 1, 0);
  p->child_edges.assign(E, 0);
  p->preorder.assign(N, 0);
  p->postorder.assign(N, 0);
  std::vector<int> marks(N, 0);
  p->topology_status = sip_lqr_compile_topology(
      E, root, edge_parents, edge_children, p->child_offsets.data(),
      p->child_edges.data(), p->parents.data(), p->children.data(),
      p->preorder.data(), p->postorder.data(), marks.data());
  *out = p;
  if (p->topology_status != SIP_LQR_SUCCESS)
    return SIP_LQR_OK; // latched, reported by factor (lqr.cpp:646-648)

  p->max_n = N ? *std::max_element(p->state_dims.begin(), p->state_dims.end()) : 0;
  p->max_m = E ? *std::max_element(p->control_dims.begin(), p->control_dims.end()) : 0;
  p->node_in.resize(N), p->node_ws.resize(N), p->node_out.resize(N);
  p->edge_in.resize(E), p->edge_ws.resize(E), p->edge_out.resize(E);
  long in = 0, ws = 0, o = 0;
  for (int i = 0; i < N; ++i) {
    const long n = p->state_dims[i];
    p->node_in[i] = in, in += n * n + 3 * n;
    p->node_out[i] = o, o += 2 * n;
  }
  for (int e = 0; e < E; ++e) {
    const long np = p->state_dims[p->parents[e]], nc = p->state_dims[p->children[e]],
               m = p->control_dims[e];
    p->edge_in[e] = in, in += nc * np + nc * m + np * m + m * m + m;
    p->edge_out[e] = o, o += m;
    p->edge_ws[e] = ws, ws += (long)p->max_n * p->max_n + m * np + m * m + m;
  }
  for (int i = 0; i < N; ++i) {
    const long n = p->state_dims[i];
    p->node_ws[i] = ws, ws += 2 * n * n + 3 * n;
  }
  p->scratch_ws = ws;
  ws += (long)p->max_m * p->max_n + (long)p->max_n * p->max_n + 2L * p->max_n + p->max_m;
  p->in_len = in, p->ws_len = ws, p->out_len = o;

  // upload the tables
  std::vector<int> ints;
  auto push_i = [&](const std::vector<int> &v) {
    const size_t at = ints.size();
    ints.insert(ints.end(), v.begin(), v.end());
    return at;
  };
  const size_t o_sd = push_i(p->state_dims), o_cd = push_i(p->control_dims),
               o_pa = push_i(p->parents), o_ch = push_i(p->children),
               o_co = push_i(p->child_offsets), o_ce = push_i(p->child_edges),
               o_pre = push_i(p->preorder), o_post = push_i(p->postorder);
  std::vector<long> longs;
  auto push_l = [&](const std::vector<long> &v) {
    const size_t at = longs.size();
    longs.insert(longs.end(), v.begin(), v.end());
    return at;
  };
  const size_t l_ni = push_l(p->node_in), l_ei = push_l(p->edge_in),
               l_nw = push_l(p->node_ws), l_ew = push_l(p->edge_ws),
               l_no = push_l(p->node_out), l_eo = push_l(p->edge_out);
  if (hipSetDevice(device) != hipSuccess ||
      hipMalloc(&p->d_ints, std::max<size_t>(1, ints.size()) * sizeof(int)) != hipSuccess ||
      hipMalloc(&p->d_longs, std::max<size_t>(1, longs.size()) * sizeof(long)) != hipSuccess ||
      hipMemcpy(p->d_ints, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->d_longs, longs.data(), longs.size() * sizeof(long), hipMemcpyHostToDevice) != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_tree_plan_create: HIP error: %s\n",
                 hipGetErrorString(hipGetLastError()));
    sip_lqr_tree_plan_destroy(p);
    *out = nullptr;
    return SIP_LQR_ERR_HIP;
  }
  const int *di = (const int *)p->d_ints;
  const long *dl = (const long *)p->d_longs;
  sipamd::tree::Meta &m = p->meta;
  m.num_edges = E, m.num_nodes = N, m.root = root, m.max_n = p->max_n, m.max_m = p->max_m;
  m.state_dims = di + o_sd, m.control_dims = di + o_cd;
  m.edge_parents = di + o_pa, m.edge_children = di + o_ch;
  m.child_offsets = di + o_co, m.child_edges = di + o_ce;
  m.preorder = di + o_pre, m.postorder = di + o_post;
  m.node_in = dl + l_ni, m.edge_in = dl + l_ei, m.node_ws = dl + l_nw;
  m.edge_ws = dl + l_ew, m.node_out = dl + l_no, m.edge_out = dl + l_eo;
  m.scratch_ws = p->scratch_ws;
  m.in_len = p->in_len, m.ws_len = p->ws_len, m.out_len = p->out_len;
  return SIP_LQR_OK;
}

void sip_lqr_tree_plan_destroy(sip_lqr_tree_plan *plan) {
  if (plan == nullptr)
    return;
  if (plan->d_ints)
    (void)hipFree(plan->d_ints);
  if (plan->d_longs)
    (void)hipFree(plan->d_longs);
  delete plan;
}

int sip_lqr_tree_topology_status(const sip_lqr_tree_plan *plan) {
  return plan ? plan->topology_status : SIP_LQR_INVALID_TOPOLOGY;
}

const int *sip_lqr_tree_topology_array(const sip_lqr_tree_plan *plan, int which) {
  if (plan == nullptr)
    return nullptr;
  switch (which) {
  case 0: return plan->child_offsets.data();
  case 1: return plan->child_edges.data();
  case 2: return plan->preorder.data();
  case 3: return plan->postorder.data();
  default: return nullptr;
  }
}

size_t sip_lqr_tree_input_len(const sip_lqr_tree_plan *p) { return p ? (size_t)p->in_len : 0; }
size_t sip_lqr_tree_work_len(const sip_lqr_tree_plan *p) { return p ? (size_t)p->ws_len : 0; }
size_t sip_lqr_tree_output_len(const sip_lqr_tree_plan *p) { return p ? (size_t)p->out_len : 0; }

size_t sip_lqr_tree_offset(const sip_lqr_tree_plan *p, int arena, int kind, int index) {
  if (p == nullptr || p->topology_status != SIP_LQR_SUCCESS || arena < 0 || arena > 2 ||
      kind < 0 || kind > 1 || index < 0 || index >= (kind == 0 ? p->num_nodes : p->num_edges))
    return (size_t)-1;
  const std::vector<long> *tab[3][2] = {{&p->node_in, &p->edge_in},
                                       {&p->node_ws, &p->edge_ws},
                                       {&p->node_out, &p->edge_out}};
  return (size_t)(*tab[arena][kind])[index];
}

int sip_lqr_tree_factor(const sip_lqr_tree_plan *plan, const double *d_input,
                        double *d_work, int32_t *d_status, void *stream) {
  if (plan == nullptr || d_status == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if (plan->topology_status != SIP_LQR_SUCCESS) {
    // every instance reports the latched traversal status (lqr.cpp:646-648)
    std::vector<int32_t> st((size_t)plan->batch, plan->topology_status);
    if (hipMemcpyAsync(d_status, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return SIP_LQR_ERR_HIP;
    return SIP_LQR_OK;
  }
  if ((d_input == nullptr && plan->in_len > 0) || d_work == nullptr)
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sipamd::tree::factor_kernel, dim3((unsigned)plan->batch),
                     dim3(sipamd::tree::TPB), 0, s, plan->meta, d_input, d_work,
                     (int *)d_status, (long)plan->batch);
  const hipError_t e = hipGetLastError();
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
      (d_output == nullptr && plan->out_len > 0) || (d_input == nullptr && plan->in_len > 0))
    return SIP_LQR_ERR_INVALID_ARGUMENT;
  if (plan->topology_status != SIP_LQR_SUCCESS)
    return SIP_LQR_ERR_INVALID_ARGUMENT; // solve() needs a successful factor
  hipLaunchKernelGGL(sipamd::tree::solve_kernel, dim3((unsigned)plan->batch),
                     dim3(sipamd::tree::TPB), 0, (hipStream_t)stream, plan->meta,
                     d_input, d_work, d_output, (const int *)d_status, (long)plan->batch);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::fprintf(stderr, "sip_lqr_tree_solve: %s\n", hipGetErrorString(e));
    return SIP_LQR_ERR_HIP;
  }
  return SIP_LQR_OK;
}

} // extern "C"
