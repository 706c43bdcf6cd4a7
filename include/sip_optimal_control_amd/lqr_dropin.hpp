// lqr_dropin.hpp -- source-compatible stand-in for the reference's
// `sip_optimal_control/lqr.hpp` (class sip::optimal_control::LQR and its
// Topology / Dimensions / Input / Output / Workspace companions,
// lqr.hpp:5-200) whose factor / solve run on an MI355X through the C ABI of
// sip_lqr_amd.h (general tree path, csrc/tree_generic.hpp).
//
// Callers written against the reference -- CallbackProvider
// (helpers.cpp:13-24, 362-368, 814-826), tests/lqr_test.cpp,
// benchmarks/lqr_benchmark.cpp -- compile against this header unchanged:
// same names, same argument meaning, same error behaviour (FactorStatus, no
// exceptions), `Input` held by reference and dereferenced at call time.
//
// Every call is a host -> device -> host round trip of one problem, so this
// is the compatibility path; throughput lives in the batched entry points.
// There is no host fallback: if no HIP device / library is available the
// process aborts with a message (the reference has no error channel for it).
//
// Header-only on top of libsip_lqr_amd.so and the HIP runtime:
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ caller.cpp -I include -I /opt/rocm/include \
//       -L sip_optimal_control_amd/lib -lsip_lqr_amd -L /opt/rocm/lib -lamdhip64
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../sip_lqr_amd.h"

namespace sip::optimal_control {

// ---- Topology (lqr.hpp:5-22; lqr.cpp:12-47) ------------------------------
struct Topology {
  int num_edges = 0;
  int root = 0;
  const int *edge_parents = nullptr;
  const int *edge_children = nullptr;

  int num_nodes() const { return num_edges + 1; }

  static constexpr int num_bytes(int edge_count) {
    return 2 * edge_count * static_cast<int>(sizeof(int));
  }
  void reserve(int edge_count) {
    num_edges = edge_count;
    edge_parents = new int[edge_count];
    edge_children = new int[edge_count];
  }
  void free() {
    delete[] edge_parents;
    delete[] edge_children;
  }
  int mem_assign(int edge_count, unsigned char *mem) {
    num_edges = edge_count;
    auto *ints = reinterpret_cast<int *>(mem);
    edge_parents = ints;
    edge_children = ints + edge_count;
    return num_bytes(edge_count);
  }
  void set_chain() { // edge e: node e -> node e+1, rooted at 0
    root = 0;
    for (int e = 0; e < num_edges; ++e) {
      const_cast<int *>(edge_parents)[e] = e;
      const_cast<int *>(edge_children)[e] = e + 1;
    }
  }
  void set_tree(int root_node, const int *parents, const int *children) {
    root = root_node;
    std::copy(parents, parents + num_edges, const_cast<int *>(edge_parents));
    std::copy(children, children + num_edges, const_cast<int *>(edge_children));
  }
};

// ---- Dimensions (lqr.hpp:24-64; lqr.cpp:49-180) ---------------------------
struct Dimensions {
  int theta_dim = 0;
  const int *state_dims = nullptr;   // per node
  const int *control_dims = nullptr; // per edge
  const int *node_c_dims = nullptr;  // optional tables: nullptr reads as 0
  const int *node_g_dims = nullptr;
  const int *edge_c_dims = nullptr;
  const int *edge_g_dims = nullptr;

  static constexpr int num_bytes(int num_edges) {
    return (3 * (num_edges + 1) + 3 * num_edges) * static_cast<int>(sizeof(int));
  }
  void reserve(int num_edges) {
    state_dims = new int[num_edges + 1];
    node_c_dims = new int[num_edges + 1];
    node_g_dims = new int[num_edges + 1];
    control_dims = new int[num_edges];
    edge_c_dims = new int[num_edges];
    edge_g_dims = new int[num_edges];
  }
  void free() {
    for (const int *t : {state_dims, control_dims, node_c_dims, node_g_dims, edge_c_dims, edge_g_dims})
      delete[] t;
  }
  int mem_assign(int num_edges, unsigned char *mem) {
    int *cur = reinterpret_cast<int *>(mem);
    auto take = [&cur](int count) { int *at = cur; cur += count; return at; };
    state_dims = take(num_edges + 1);
    control_dims = take(num_edges);
    node_c_dims = take(num_edges + 1);
    node_g_dims = take(num_edges + 1);
    edge_c_dims = take(num_edges);
    edge_g_dims = take(num_edges);
    return num_bytes(num_edges);
  }
  void set_uniform(int num_edges, int state_dim, int control_dim, int node_c_dim, int node_g_dim,
                   int edge_c_dim, int edge_g_dim, int global_dim = 0) {
    theta_dim = global_dim;
    auto fill = [](const int *t, int count, int v) { std::fill_n(const_cast<int *>(t), count, v); };
    fill(state_dims, num_edges + 1, state_dim);
    fill(node_c_dims, num_edges + 1, node_c_dim);
    fill(node_g_dims, num_edges + 1, node_g_dim);
    fill(control_dims, num_edges, control_dim);
    fill(edge_c_dims, num_edges, edge_c_dim);
    fill(edge_g_dims, num_edges, edge_g_dim);
  }

  int get_schur_dim() const { return theta_dim; }
  int get_state_dim(int node) const { return state_dims[node]; }
  int get_control_dim(int edge) const { return control_dims[edge]; }
  int get_node_c_dim(int node) const { return node_c_dims ? node_c_dims[node] : 0; }
  int get_node_g_dim(int node) const { return node_g_dims ? node_g_dims[node] : 0; }
  int get_edge_c_dim(int edge) const { return edge_c_dims ? edge_c_dims[edge] : 0; }
  int get_edge_g_dim(int edge) const { return edge_g_dims ? edge_g_dims[edge] : 0; }
  int max_state_dim(int num_nodes) const { return largest(state_dims, num_nodes); }
  int max_control_dim(int num_edges) const { return largest(control_dims, num_edges); }
  int max_node_c_dim(int num_nodes) const { return largest(node_c_dims, num_nodes); }
  int max_node_g_dim(int num_nodes) const { return largest(node_g_dims, num_nodes); }
  int max_edge_c_dim(int num_edges) const { return largest(edge_c_dims, num_edges); }
  int max_edge_g_dim(int num_edges) const { return largest(edge_g_dims, num_edges); }
  int get_stagewise_x_dim(int num_edges) const {
    int total = state_dims[num_edges];
    for (int e = 0; e < num_edges; ++e)
      total += state_dims[e] + control_dims[e];
    return total;
  }
  int get_x_dim(int num_edges) const { return get_stagewise_x_dim(num_edges) + theta_dim; }
  int get_y_dim(int num_edges) const {
    int total = 0;
    for (int node = 0; node <= num_edges; ++node)
      total += state_dims[node] + get_node_c_dim(node);
    for (int e = 0; e < num_edges; ++e)
      total += get_edge_c_dim(e);
    return total;
  }
  int get_z_dim(int num_edges) const {
    int total = 0;
    for (int node = 0; node <= num_edges; ++node)
      total += get_node_g_dim(node);
    for (int e = 0; e < num_edges; ++e)
      total += get_edge_g_dim(e);
    return total;
  }
  int get_stagewise_kkt_dim(int num_edges) const {
    return get_stagewise_x_dim(num_edges) + get_y_dim(num_edges) + get_z_dim(num_edges);
  }

private:
  static int largest(const int *t, int count) {
    return (t == nullptr || count == 0) ? 0 : *std::max_element(t, t + count);
  }
};

// ---- LQR (lqr.hpp:66-200) --------------------------------------------------
class LQR {
public:
  enum class FactorStatus {
    SUCCESS = SIP_LQR_SUCCESS,
    INVALID_DELTA = SIP_LQR_INVALID_DELTA,
    F_FACTORIZATION_FAILURE = SIP_LQR_F_FACTORIZATION_FAILURE,
    G_FACTORIZATION_FAILURE = SIP_LQR_G_FACTORIZATION_FAILURE,
    INVALID_TOPOLOGY = SIP_LQR_INVALID_TOPOLOGY,
  };

  struct Input {
    double **Q, **M, **R, **q, **r, **A, **B, **c, **delta;
    const Dimensions &dimensions;
    const Topology &topology;
  };

  struct Output {
    double **x, **u, **y;
    static constexpr auto num_bytes(int num_edges) -> int {
      return (3 * num_edges + 2) * static_cast<int>(sizeof(double *));
    }
    void reserve(int num_edges) {
      x = new double *[num_edges + 1];
      y = new double *[num_edges + 1];
      u = new double *[num_edges];
    }
    void free() {
      delete[] x;
      delete[] u;
      delete[] y;
    }
    auto mem_assign(int num_edges, unsigned char *mem) -> int {
      auto **tab = reinterpret_cast<double **>(mem);
      x = tab;
      u = tab + (num_edges + 1);
      y = tab + (2 * num_edges + 1);
      return num_bytes(num_edges);
    }
  };

  // Same fields and meaning as the reference's workspace (lqr.hpp:109-135).
  // One routine (`walk`) enumerates every table, block and int array with its
  // size; reserve / mem_assign / num_bytes are three visitors over that walk,
  // so the three cannot drift apart (the reference asserts their equality,
  // lqr.cpp:431).
  struct Workspace {
    double **W, **K, **V, **G_factor, **F_factor, **sqrt_delta, **sqrt_delta_inv, **k, **v;
    double *G, *g, *H, *h, *F, *f;
    int *child_offsets, *child_edges, *edge_parents, *edge_children;
    int *preorder_nodes, *postorder_nodes, *node_marks;

    void reserve(int state_dim, int control_dim, int num_edges) {
      UniformShape s(state_dim, control_dim, num_edges);
      reserve(s.dims, s.topo);
    }
    void reserve(const Dimensions &dims, const Topology &topo) {
      walk(dims, topo, HeapVisitor{});
    }
    void free(int num_edges) {
      for (int e = 0; e < num_edges; ++e)
        for (double **tab : {W, K, G_factor, k})
          delete[] tab[e];
      for (int node = 0; node <= num_edges; ++node)
        for (double **tab : {V, F_factor, sqrt_delta, sqrt_delta_inv, v})
          delete[] tab[node];
      for (double **tab : {W, K, V, G_factor, F_factor, sqrt_delta, sqrt_delta_inv, k, v})
        delete[] tab;
      for (double *blk : {G, g, H, h, F, f})
        delete[] blk;
      for (int *arr : {child_offsets, child_edges, edge_parents, edge_children, preorder_nodes,
                       postorder_nodes, node_marks})
        delete[] arr;
    }
    auto mem_assign(const Dimensions &dims, const Topology &topo, unsigned char *mem) -> int {
      ArenaVisitor a{mem, 0};
      walk(dims, topo, a);
      return a.used;
    }
    static constexpr auto num_bytes(int state_dim, int control_dim, int num_edges) -> int {
      const int n = state_dim, m = control_dim, T = num_edges, N = num_edges + 1;
      const int d = static_cast<int>(sizeof(double)), p = static_cast<int>(sizeof(double *)),
                i = static_cast<int>(sizeof(int));
      return (4 * T + 5 * N) * p                                    // pointer tables
             + T * (n * n + m * n + m * m + m) * d                  // W K G_factor k
             + N * (2 * n * n + 3 * n) * d                          // V F_factor sd sdi v
             + (m * m + 2 * n + m * n + m + n * n) * d              // G g H h F f
             + ((N + 1) + 3 * T + 3 * N) * i;                       // traversal arrays
    }
    static auto num_bytes(const Dimensions &dims, const Topology &topo) -> int {
      ArenaVisitor a{nullptr, 0};
      Workspace scratch{};
      scratch.walk(dims, topo, a);
      return a.used;
    }

  private:
    struct UniformShape {
      Dimensions dims;
      Topology topo;
      std::vector<int> sd, cd, pa, ch;
      UniformShape(int n, int m, int T) : sd(T + 1, n), cd(T, m), pa(T), ch(T) {
        for (int e = 0; e < T; ++e)
          pa[e] = e, ch[e] = e + 1;
        dims.state_dims = sd.data();
        dims.control_dims = cd.data();
        topo.num_edges = T;
        topo.edge_parents = pa.data();
        topo.edge_children = ch.data();
      }
    };
    struct HeapVisitor {
      void table(double **&t, int count) { t = new double *[count > 0 ? count : 1]; }
      void block(double *&b, int count) { b = new double[count > 0 ? count : 1]; }
      void ints(int *&a, int count) { a = new int[count > 0 ? count : 1]; }
    };
    struct ArenaVisitor { // mem == nullptr: size only
      unsigned char *mem;
      int used;
      template <class T> void carve(T *&ptr, int count) {
        if (mem != nullptr)
          ptr = reinterpret_cast<T *>(mem + used);
        used += count * static_cast<int>(sizeof(T));
      }
      void table(double **&t, int count) { carve(t, count); }
      void block(double *&b, int count) { carve(b, count); }
      void ints(int *&a, int count) { carve(a, count); }
    };
    // Order and slot sizes follow lqr.cpp:321-434 (W slots are max_n^2, K slots
    // m_e x max_n), so byte counts agree with the reference's num_bytes.
    template <class Visitor> void walk(const Dimensions &dims, const Topology &topo, Visitor &&vis) {
      const int E = topo.num_edges, N = topo.num_nodes();
      const int max_n = dims.max_state_dim(N), max_m = dims.max_control_dim(E);
      const bool assign = true;
      (void)assign;
      vis.table(W, E), vis.table(K, E), vis.table(V, N), vis.table(G_factor, E), vis.table(F_factor, N);
      vis.table(sqrt_delta, N), vis.table(sqrt_delta_inv, N), vis.table(k, E), vis.table(v, N);
      double *unused = nullptr;
      auto slot = [&](double **tab, int index, int count) {
        if (tab != nullptr)
          vis.block(tab[index], count);
        else
          vis.block(unused, count);
      };
      for (int e = 0; e < E; ++e) {
        const int m = dims.get_control_dim(e);
        slot(W, e, max_n * max_n), slot(K, e, m * max_n), slot(G_factor, e, m * m), slot(k, e, m);
      }
      for (int node = 0; node < N; ++node) {
        const int n = dims.get_state_dim(node);
        slot(V, node, n * n), slot(F_factor, node, n * n), slot(sqrt_delta, node, n);
        slot(sqrt_delta_inv, node, n), slot(v, node, n);
      }
      vis.block(G, max_m * max_m), vis.block(g, max_n), vis.block(H, max_m * max_n);
      vis.block(h, max_m), vis.block(F, max_n * max_n), vis.block(f, max_n);
      vis.ints(child_offsets, N + 1), vis.ints(child_edges, E), vis.ints(edge_parents, E);
      vis.ints(edge_children, E), vis.ints(preorder_nodes, N), vis.ints(postorder_nodes, N);
      vis.ints(node_marks, N);
    }
  };

  LQR(const Input &data, Workspace &workspace)
      : input_(data), workspace_(workspace), traversal_status_(FactorStatus::INVALID_TOPOLOGY) {
    compile_topology();
  }
  LQR(LQR &&) = default;

  // ---- additions of this adapter (the reference has neither devices nor kernels to choose) ----
  // HIP device ordinal of the objects constructed afterwards (default: SIP_LQR_DROPIN_DEVICE of the
  // environment, else 0) / of this object (before its first factor or solve).
  static int &default_device() {
    static int dev = [] {
      const char *v = std::getenv("SIP_LQR_DROPIN_DEVICE");
      return v != nullptr ? std::atoi(v) : 0;
    }();
    return dev;
  }
  void set_device(int ordinal) {
    device_ordinal_ = ordinal;
    gpu_.reset();
    fast_.reset();
  }
  // Uniform chains (Topology::set_chain + Dimensions::set_uniform) with a fused kernel (n <= 16,
  // m <= 8) on the fused fp64 kernels of the batched path (sip_lqr_factor / sip_lqr_solve, batch 1)
  // instead of the general tree engine: about an order of magnitude less latency per call.  The price:
  // of the public LQR::Workspace fields only K and k are filled -- W, V, G_factor, F_factor,
  // sqrt_delta(_inv) and v (read by helpers.cpp:521-665, the multi-rhs solve) stay untouched -- so it
  // is opt-in (also: SIP_LQR_DROPIN_FUSED=1 in the environment).  Other problems ignore the switch.
  static bool &default_fused_chains() {
    static bool on = [] {
      const char *v = std::getenv("SIP_LQR_DROPIN_FUSED");
      return v != nullptr && v[0] == '1';
    }();
    return on;
  }
  void set_fused_chains(bool on) { fused_chains_ = on; }
  // Trees and chains run on the fused size-class kernel (csrc/tree_qw16.hpp, every LQR::Workspace field written
  // out) wherever it applies; true keeps them on the general engine (also: SIP_LQR_DROPIN_GENERAL=1).
  static bool &default_general_engine() {
    static bool on = [] {
      const char *v = std::getenv("SIP_LQR_DROPIN_GENERAL");
      return v != nullptr && v[0] == '1';
    }();
    return on;
  }
  void set_general_engine(bool on) {
    general_engine_ = on;
    gpu_.reset();
  }
  bool uses_fused_tree_kernel() {
    OnDevice on_device(device_ordinal_);
    return traversal_status_ == FactorStatus::SUCCESS && fast_chain() == nullptr && device().d_scratch != nullptr;
  }
  bool uses_fused_chain_kernel() {
    OnDevice on_device(device_ordinal_);
    return fast_chain() != nullptr;
  }

  // Replaces compile_topology_data (lqr.cpp:563-631); fills the traversal
  // arrays of the caller's workspace (read by helpers.cpp:217-218, 521-665 and
  // tests/lqr_test.cpp:940-950) and (re)creates the device plan.
  auto compile_topology() -> FactorStatus {
    const Topology &t = input_.topology;
    const int st = sip_lqr_compile_topology(
        t.num_edges, t.root, t.edge_parents, t.edge_children, workspace_.child_offsets,
        workspace_.child_edges, workspace_.edge_parents, workspace_.edge_children,
        workspace_.preorder_nodes, workspace_.postorder_nodes, workspace_.node_marks);
    traversal_status_ = static_cast<FactorStatus>(st);
    gpu_.reset();
    fast_.reset();
    return traversal_status_;
  }

  // Replaces lqr.cpp:645-731.
  FactorStatus factor_with_status() {
    if (traversal_status_ != FactorStatus::SUCCESS)
      return traversal_status_;
    OnDevice on_device(device_ordinal_); // copies and null-stream kernels on the plan's device; caller's restored
    if (FastChain *f = fast_chain())
      return fast_factor(*f);
    Device &d = device();
    gather_input(d);
    d.h2d(d.d_in, d.h_in);
    if (d.d_scratch != nullptr) // the fused size-class kernel, writing every LQR::Workspace field (tree_qw16.hpp)
      check(sip_lqr_tree_factor_solve_workspace(d.plan, d.d_in, d.d_ws, d.d_out, d.d_status, d.d_scratch, nullptr),
            "sip_lqr_tree_factor_solve_workspace");
    else
      check(sip_lqr_tree_factor(d.plan, d.d_in, d.d_ws, d.d_status, nullptr), "sip_lqr_tree_factor");
    int32_t st = 0;
    check_hip(hipMemcpy(&st, d.d_status, sizeof(st), hipMemcpyDeviceToHost), "status copy");
    d.d2h(d.h_ws, d.d_ws);
    scatter_factor_state(d);
    return static_cast<FactorStatus>(st);
  }
  bool factor() { return factor_with_status() == FactorStatus::SUCCESS; }

  // Replaces lqr.cpp:735-871; requires a preceding successful factor.
  void solve(Output &output) {
    OnDevice on_device(device_ordinal_);
    if (FastChain *f = fast_chain())
      return fast_solve(*f, output);
    Device &d = device();
    gather_input(d);
    d.h2d(d.d_in, d.h_in);
    if (d.d_scratch != nullptr) // one fused sweep (it refactors: same matrices, same factor state)
      check(sip_lqr_tree_factor_solve_workspace(d.plan, d.d_in, d.d_ws, d.d_out, d.d_status, d.d_scratch, nullptr),
            "sip_lqr_tree_factor_solve_workspace");
    else
      check(sip_lqr_tree_solve(d.plan, d.d_in, d.d_ws, d.d_out, d.d_status, nullptr), "sip_lqr_tree_solve");
    d.d2h(d.h_out, d.d_out);
    d.d2h(d.h_ws, d.d_ws);
    const Dimensions &dims = input_.dimensions;
    const int E = input_.topology.num_edges;
    for (int node = 0; node <= E; ++node) {
      const int n = dims.get_state_dim(node);
      const double *src = d.h_out.data() + sip_lqr_tree_offset(d.plan, 2, 0, node);
      std::copy(src, src + n, output.x[node]);
      std::copy(src + n, src + 2 * n, output.y[node]);
      const double *ws = d.h_ws.data() + sip_lqr_tree_offset(d.plan, 1, 0, node);
      std::copy(ws + 2L * n * n + 2 * n, ws + 2L * n * n + 3 * n, workspace_.v[node]);
    }
    for (int e = 0; e < E; ++e) {
      const int m = dims.get_control_dim(e);
      const int np = dims.get_state_dim(workspace_.edge_parents[e]);
      const double *src = d.h_out.data() + sip_lqr_tree_offset(d.plan, 2, 1, e);
      std::copy(src, src + m, output.u[e]);
      const double *ws = d.h_ws.data() + sip_lqr_tree_offset(d.plan, 1, 1, e) + (long)d.max_n * d.max_n +
                         (long)m * np + (long)m * m;
      std::copy(ws, ws + m, workspace_.k[e]);
    }
  }

private:
  // Make the object's device current for the duration of a call and restore the caller's on return: the
  // synchronous copies and the plan's null-stream kernels then share one device's null stream, whatever
  // device the host program had current (a multi-GPU host switches between calls).
  struct OnDevice {
    int previous = -1;
    explicit OnDevice(int ordinal) {
      if (hipGetDevice(&previous) != hipSuccess)
        previous = -1;
      if (previous != ordinal)
        check_hip(hipSetDevice(ordinal), "hipSetDevice");
      else
        previous = -1;
    }
    ~OnDevice() {
      if (previous >= 0)
        (void)hipSetDevice(previous);
    }
    OnDevice(const OnDevice &) = delete;
    OnDevice &operator=(const OnDevice &) = delete;
  };

  // Device-side state of one LQR object (plan + arenas for a batch of one).
  struct Device {
    sip_lqr_tree_plan *plan = nullptr;
    double *d_in = nullptr, *d_ws = nullptr, *d_out = nullptr;
    void *d_scratch = nullptr; // non-null: the plan runs on the fused tree kernel
    int32_t *d_status = nullptr;
    std::vector<double> h_in, h_ws, h_out;
    int max_n = 0;
    void h2d(double *dst, const std::vector<double> &src) {
      if (!src.empty())
        check_hip(hipMemcpy(dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice), "H2D");
    }
    void d2h(std::vector<double> &dst, const double *src) {
      if (!dst.empty())
        check_hip(hipMemcpy(dst.data(), src, dst.size() * sizeof(double), hipMemcpyDeviceToHost), "D2H");
    }
    ~Device() {
      for (void *p : {(void *)d_in, (void *)d_ws, (void *)d_out, (void *)d_status, d_scratch})
        if (p != nullptr)
          (void)hipFree(p);
      sip_lqr_tree_plan_destroy(plan);
    }
  };

  static void check(int code, const char *what) {
    if (code != SIP_LQR_OK) {
      std::fprintf(stderr, "sip::optimal_control::LQR (MI355X): %s failed with code %d; no host fallback\n", what,
                   code);
      std::abort();
    }
  }
  static void check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) {
      std::fprintf(stderr, "sip::optimal_control::LQR (MI355X): %s: %s; no host fallback\n", what,
                   hipGetErrorString(e));
      std::abort();
    }
  }

  // Fused-kernel state of a uniform chain (batch 1, packed chain layout of sip_lqr_amd.h).
  struct FastChain {
    sip_lqr_plan *plan = nullptr;
    int n = 0, m = 0, T = 0;
    void *d_mats = nullptr, *d_vecs = nullptr, *d_sol = nullptr, *d_gains = nullptr, *d_ws = nullptr;
    int32_t *d_status = nullptr;
    std::vector<double> h_mats, h_vecs, h_sol, h_gains;
    ~FastChain() {
      for (void *p : {d_mats, d_vecs, d_sol, d_gains, d_ws, (void *)d_status})
        if (p != nullptr)
          (void)hipFree(p);
      sip_lqr_plan_destroy(plan);
    }
  };
  // non-null iff the switch is on and the problem is a uniform chain with a fused fp64 kernel
  FastChain *fast_chain() {
    if (!fused_chains_)
      return nullptr;
    if (fast_)
      return fast_->plan != nullptr ? fast_.get() : nullptr;
    fast_ = std::make_unique<FastChain>(); // plan == nullptr: "looked, not applicable"
    const Topology &t = input_.topology;
    const Dimensions &dims = input_.dimensions;
    const int E = t.num_edges;
    if (E < 1 || t.root != 0)
      return nullptr;
    const int n = dims.get_state_dim(0), m = dims.get_control_dim(0);
    if (n < 1 || m < 1 || n > 16 || m > 8)
      return nullptr;
    for (int e = 0; e < E; ++e)
      if (t.edge_parents[e] != e || t.edge_children[e] != e + 1 || dims.get_control_dim(e) != m ||
          dims.get_state_dim(e + 1) != n)
        return nullptr;
    FastChain &f = *fast_;
    f.n = n, f.m = m, f.T = E;
    check(sip_lqr_plan_create(SIP_LQR_F64, 1, E, n, m, device_ordinal_, &f.plan), "sip_lqr_plan_create");
    f.h_mats.assign(sip_lqr_mats_len(f.plan), 0.0), f.h_vecs.assign(sip_lqr_vecs_len(f.plan), 0.0);
    f.h_sol.assign(sip_lqr_vecs_len(f.plan), 0.0), f.h_gains.assign(sip_lqr_gains_len(f.plan), 0.0);
    auto alloc = [](void *&p, size_t bytes) { check_hip(hipMalloc(&p, std::max<size_t>(16, bytes)), "hipMalloc"); };
    alloc(f.d_mats, sip_lqr_mats_bytes(f.plan)), alloc(f.d_vecs, sip_lqr_vecs_bytes(f.plan));
    alloc(f.d_sol, sip_lqr_sol_bytes(f.plan)), alloc(f.d_gains, sip_lqr_gains_bytes(f.plan));
    alloc(f.d_ws, sip_lqr_workspace_bytes(f.plan));
    check_hip(hipMalloc((void **)&f.d_status, sizeof(int32_t)), "hipMalloc");
    return &f;
  }
  // null tables (not yet patched by the caller, helpers.cpp:362-367 / 814-816) are passed as zeros
  void fast_pack(FastChain &f) {
    const int N = f.T + 1;
    std::vector<double> zero((size_t)std::max(f.n * f.n, f.n * f.m) + 1, 0.0);
    auto table = [&](double *const *tab, int count) {
      std::vector<double *> out((size_t)count);
      for (int i = 0; i < count; ++i)
        out[i] = (tab != nullptr && tab[i] != nullptr) ? tab[i] : zero.data();
      return out;
    };
    auto Q = table(input_.Q, N), q = table(input_.q, N), c = table(input_.c, N), dl = table(input_.delta, N);
    auto Mx = table(input_.M, f.T), R = table(input_.R, f.T), r = table(input_.r, f.T), A = table(input_.A, f.T),
         B = table(input_.B, f.T);
    check(sip_lqr_pack_problem(f.plan, 0, Q.data(), Mx.data(), R.data(), q.data(), r.data(), A.data(), B.data(),
                               c.data(), dl.data(), f.h_mats.data(), f.h_vecs.data()),
          "sip_lqr_pack_problem");
  }
  FactorStatus fast_factor(FastChain &f) {
    fast_pack(f);
    check_hip(hipMemcpy(f.d_mats, f.h_mats.data(), f.h_mats.size() * sizeof(double), hipMemcpyHostToDevice), "H2D");
    check(sip_lqr_factor(f.plan, f.d_mats, f.d_gains, f.d_status, f.d_ws, nullptr), "sip_lqr_factor");
    int32_t st = 0;
    check_hip(hipMemcpy(&st, f.d_status, sizeof(st), hipMemcpyDeviceToHost), "status copy");
    fast_gains(f);
    return static_cast<FactorStatus>(st);
  }
  void fast_gains(FastChain &f) { // K (after factor) and k (after solve) into the caller's workspace
    check_hip(hipMemcpy(f.h_gains.data(), f.d_gains, f.h_gains.size() * sizeof(double), hipMemcpyDeviceToHost), "D2H");
    if (workspace_.K != nullptr && workspace_.k != nullptr)
      check(sip_lqr_unpack_gains(f.plan, 0, f.h_gains.data(), workspace_.K, workspace_.k), "sip_lqr_unpack_gains");
  }
  void fast_solve(FastChain &f, Output &output) {
    fast_pack(f); // mats too: the caller may have re-patched the tables since factor (same values then)
    check_hip(hipMemcpy(f.d_vecs, f.h_vecs.data(), f.h_vecs.size() * sizeof(double), hipMemcpyHostToDevice), "H2D");
    check(sip_lqr_solve(f.plan, f.d_mats, f.d_vecs, f.d_sol, f.d_gains, f.d_ws, nullptr), "sip_lqr_solve");
    check_hip(hipMemcpy(f.h_sol.data(), f.d_sol, f.h_sol.size() * sizeof(double), hipMemcpyDeviceToHost), "D2H");
    check(sip_lqr_unpack_solution(f.plan, 0, f.h_sol.data(), output.x, output.u, output.y), "sip_lqr_unpack_solution");
    fast_gains(f);
  }

  Device &device() {
    if (gpu_)
      return *gpu_;
    auto d = std::make_unique<Device>();
    const Topology &t = input_.topology;
    const Dimensions &dims = input_.dimensions;
    check(sip_lqr_tree_plan_create(1, t.num_edges, t.root, t.edge_parents, t.edge_children, dims.state_dims,
                                   dims.control_dims, device_ordinal_, &d->plan),
          "sip_lqr_tree_plan_create");
    d->max_n = dims.max_state_dim(t.num_nodes());
    d->h_in.assign(sip_lqr_tree_input_len(d->plan), 0.0);
    d->h_ws.assign(sip_lqr_tree_work_len(d->plan), 0.0);
    d->h_out.assign(sip_lqr_tree_output_len(d->plan), 0.0);
    auto dev_alloc = [](size_t count) {
      void *p = nullptr;
      check_hip(hipMalloc(&p, std::max<size_t>(1, count) * sizeof(double)), "hipMalloc");
      return static_cast<double *>(p);
    };
    d->d_in = dev_alloc(d->h_in.size());
    d->d_ws = dev_alloc(d->h_ws.size());
    d->d_out = dev_alloc(d->h_out.size());
    check_hip(hipMalloc((void **)&d->d_status, sizeof(int32_t)), "hipMalloc");
    check_hip(hipMemset(d->d_status, 0xff, sizeof(int32_t)), "hipMemset");
    // Default: the fused size-class kernel (state dims <= 15, control dims <= 8) with every LQR::Workspace field
    // written out; set_general_engine(true) / SIP_LQR_DROPIN_GENERAL=1 keeps the general engine.
    const size_t scratch = general_engine_ ? 0 : sip_lqr_tree_fused_scratch_bytes(d->plan);
    if (scratch > 0)
      check_hip(hipMalloc(&d->d_scratch, scratch), "hipMalloc");
    gpu_ = std::move(d);
    return *gpu_;
  }

  // Input tables are dereferenced now, not at construction (helpers.cpp:362-367
  // patches them right before the call).  Null tables (not yet patched) are
  // skipped.
  void gather_input(Device &d) {
    const Dimensions &dims = input_.dimensions;
    const int E = input_.topology.num_edges;
    auto put = [](double *dst, double *const *tab, int index, long count) {
      if (tab != nullptr && tab[index] != nullptr && count > 0)
        std::copy(tab[index], tab[index] + count, dst);
      return dst + count;
    };
    for (int node = 0; node <= E; ++node) {
      const long n = dims.get_state_dim(node);
      double *dst = d.h_in.data() + sip_lqr_tree_offset(d.plan, 0, 0, node);
      dst = put(dst, input_.Q, node, n * n);
      dst = put(dst, input_.q, node, n);
      dst = put(dst, input_.c, node, n);
      dst = put(dst, input_.delta, node, n);
    }
    for (int e = 0; e < E; ++e) {
      const long np = dims.get_state_dim(workspace_.edge_parents[e]);
      const long nc = dims.get_state_dim(workspace_.edge_children[e]);
      const long m = dims.get_control_dim(e);
      double *dst = d.h_in.data() + sip_lqr_tree_offset(d.plan, 0, 1, e);
      dst = put(dst, input_.A, e, nc * np);
      dst = put(dst, input_.B, e, nc * m);
      dst = put(dst, input_.M, e, np * m);
      dst = put(dst, input_.R, e, m * m);
      dst = put(dst, input_.r, e, m);
    }
  }

  // Device factor state -> the caller's LQR::Workspace fields.
  void scatter_factor_state(Device &d) {
    const Dimensions &dims = input_.dimensions;
    const int E = input_.topology.num_edges;
    for (int e = 0; e < E; ++e) {
      const long np = dims.get_state_dim(workspace_.edge_parents[e]);
      const long nc = dims.get_state_dim(workspace_.edge_children[e]);
      const long m = dims.get_control_dim(e);
      const double *src = d.h_ws.data() + sip_lqr_tree_offset(d.plan, 1, 1, e);
      std::copy(src, src + nc * nc, workspace_.W[e]);
      src += (long)d.max_n * d.max_n;
      std::copy(src, src + m * np, workspace_.K[e]);
      src += m * np;
      std::copy(src, src + m * m, workspace_.G_factor[e]);
    }
    for (int node = 0; node <= E; ++node) {
      const long n = dims.get_state_dim(node);
      const double *src = d.h_ws.data() + sip_lqr_tree_offset(d.plan, 1, 0, node);
      std::copy(src, src + n * n, workspace_.V[node]);
      std::copy(src + n * n, src + 2 * n * n, workspace_.F_factor[node]);
      std::copy(src + 2 * n * n, src + 2 * n * n + n, workspace_.sqrt_delta[node]);
      std::copy(src + 2 * n * n + n, src + 2 * n * n + 2 * n, workspace_.sqrt_delta_inv[node]);
    }
  }

  const Input &input_;
  Workspace &workspace_;
  FactorStatus traversal_status_;
  std::unique_ptr<Device> gpu_;
  std::unique_ptr<FastChain> fast_;
  int device_ordinal_ = default_device();
  bool fused_chains_ = default_fused_chains();
  bool general_engine_ = default_general_engine();
};

} // namespace sip::optimal_control
