// callback_provider_dropin.hpp -- source-level drop-in for the reference's
// `sip::optimal_control::CallbackProvider` (helpers.hpp:7-33) and the parts of
// `types.hpp` it touches, over the batched C ABI of sip_kkt_amd.h (batch = 1).
//
//   CallbackProvider(input, workspace)            helpers.cpp:11-26
//   bool factor(w, r1, r2, r3)                    helpers.cpp:242-407
//   void solve(b, sol)                            helpers.cpp:896-951
//   void add_Kx_to_y(w, r1, r2, r3, x_x, x_y, x_z, y_x, y_y, y_z)   helpers.cpp:953-976
//
// Same member names and argument meaning; the model-callback outputs are the
// reference's pointer-per-field structs (types.hpp:48-89).  Every call is a
// host -> device -> host round trip of one problem (the compatibility path;
// the throughput path is the batched C ABI with device-resident arenas).
//
// Deliberately smaller than types.hpp: `Input` carries only `dimensions` and
// `topology` (what CallbackProvider reads, helpers.cpp:11-26), `Workspace`
// only `model_callback_output` and the flattened-ordering metadata
// (types.cpp:24-64) -- no SIP front-end workspace (`::sip::Workspace`,
// `::sip::Settings` are not part of this repo), so `Workspace::reserve` takes
// (dimensions, topology) only.  No GPU / no library: the process aborts with a
// message, like lqr_dropin.hpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sip_kkt_amd.h"
#include "lqr_dropin.hpp"

namespace sip::optimal_control {

// types.hpp:48-62
struct NodeModelCallbackOutput {
  double f = 0.0;
  double *df_dx = nullptr, *df_dtheta = nullptr;
  double *c = nullptr, *dc_dx = nullptr, *dc_dtheta = nullptr;
  double *g = nullptr, *dg_dx = nullptr, *dg_dtheta = nullptr;
  double *d2L_dx2 = nullptr, *d2L_dxdtheta = nullptr, *d2L_dtheta2 = nullptr;
};

// types.hpp:66-89
struct EdgeModelCallbackOutput {
  double f = 0.0;
  double *df_dx = nullptr, *df_du = nullptr, *df_dtheta = nullptr;
  double *dyn_res = nullptr, *ddyn_dx = nullptr, *ddyn_du = nullptr, *ddyn_dtheta = nullptr;
  double *c = nullptr, *dc_dx = nullptr, *dc_du = nullptr, *dc_dtheta = nullptr;
  double *g = nullptr, *dg_dx = nullptr, *dg_du = nullptr, *dg_dtheta = nullptr;
  double *d2L_dx2 = nullptr, *d2L_dxdu = nullptr, *d2L_du2 = nullptr;
  double *d2L_dxdtheta = nullptr, *d2L_dudtheta = nullptr, *d2L_dtheta2 = nullptr;
};

// types.hpp:91-126 (reserve / free; one heap block per field, like types.cpp)
struct ModelCallbackOutput {
  NodeModelCallbackOutput *nodes = nullptr;
  EdgeModelCallbackOutput *edges = nullptr;

  void reserve(const Dimensions &dim, const Topology &topology) {
    const int E = topology.num_edges, N = E + 1, p = dim.theta_dim;
    auto block = [](int count) { return new double[count > 0 ? count : 1](); };
    nodes = new NodeModelCallbackOutput[N];
    edges = new EdgeModelCallbackOutput[E > 0 ? E : 1];
    for (int i = 0; i < N; ++i) {
      const int n = dim.get_state_dim(i), c = dim.get_node_c_dim(i), g = dim.get_node_g_dim(i);
      auto &o = nodes[i];
      o.df_dx = block(n), o.df_dtheta = block(p), o.c = block(c), o.dc_dx = block(c * n);
      o.dc_dtheta = block(c * p), o.g = block(g), o.dg_dx = block(g * n), o.dg_dtheta = block(g * p);
      o.d2L_dx2 = block(n * n), o.d2L_dxdtheta = block(n * p), o.d2L_dtheta2 = block(p * p);
    }
    for (int e = 0; e < E; ++e) {
      const int n = dim.get_state_dim(topology.edge_parents[e]), nc = dim.get_state_dim(topology.edge_children[e]);
      const int m = dim.get_control_dim(e), c = dim.get_edge_c_dim(e), g = dim.get_edge_g_dim(e);
      auto &o = edges[e];
      o.df_dx = block(n), o.df_du = block(m), o.df_dtheta = block(p), o.dyn_res = block(nc);
      o.ddyn_dx = block(nc * n), o.ddyn_du = block(nc * m), o.ddyn_dtheta = block(nc * p);
      o.c = block(c), o.dc_dx = block(c * n), o.dc_du = block(c * m), o.dc_dtheta = block(c * p);
      o.g = block(g), o.dg_dx = block(g * n), o.dg_du = block(g * m), o.dg_dtheta = block(g * p);
      o.d2L_dx2 = block(n * n), o.d2L_dxdu = block(n * m), o.d2L_du2 = block(m * m);
      o.d2L_dxdtheta = block(n * p), o.d2L_dudtheta = block(m * p), o.d2L_dtheta2 = block(p * p);
    }
  }
  void free(const Topology &topology) {
    const int E = topology.num_edges, N = E + 1;
    for (int i = 0; i < N; ++i) {
      auto &o = nodes[i];
      for (double *b : {o.df_dx, o.df_dtheta, o.c, o.dc_dx, o.dc_dtheta, o.g, o.dg_dx, o.dg_dtheta, o.d2L_dx2,
                        o.d2L_dxdtheta, o.d2L_dtheta2})
        delete[] b;
    }
    for (int e = 0; e < E; ++e) {
      auto &o = edges[e];
      for (double *b : {o.df_dx, o.df_du, o.df_dtheta, o.dyn_res, o.ddyn_dx, o.ddyn_du, o.ddyn_dtheta, o.c, o.dc_dx,
                        o.dc_du, o.dc_dtheta, o.g, o.dg_dx, o.dg_du, o.dg_dtheta, o.d2L_dx2, o.d2L_dxdu, o.d2L_du2,
                        o.d2L_dxdtheta, o.d2L_dudtheta, o.d2L_dtheta2})
        delete[] b;
    }
    delete[] nodes;
    delete[] edges;
    nodes = nullptr, edges = nullptr;
  }
};

// types.hpp:128-156, the members CallbackProvider reads
struct Input {
  Dimensions dimensions;
  Topology topology;
};

// types.hpp:162-322, the members CallbackProvider and its callers read
struct Workspace {
  ModelCallbackOutput model_callback_output;
  int stagewise_x_dim = 0, x_dim = 0, y_dim = 0, z_dim = 0, stagewise_kkt_dim = 0;
  int *x_state_offsets = nullptr, *x_control_offsets = nullptr, *y_dyn_offsets = nullptr,
      *y_node_c_offsets = nullptr, *y_edge_c_offsets = nullptr, *z_node_offsets = nullptr,
      *z_edge_offsets = nullptr;

  void reserve(const Dimensions &dim, const Topology &topology) {
    const int E = topology.num_edges, N = E + 1;
    model_callback_output.reserve(dim, topology);
    x_state_offsets = new int[N], y_dyn_offsets = new int[N], y_node_c_offsets = new int[N];
    z_node_offsets = new int[N];
    x_control_offsets = new int[E > 0 ? E : 1], y_edge_c_offsets = new int[E > 0 ? E : 1];
    z_edge_offsets = new int[E > 0 ? E : 1];
    // populate_workspace_metadata, types.cpp:24-64
    stagewise_x_dim = dim.get_stagewise_x_dim(E), x_dim = dim.get_x_dim(E), y_dim = dim.get_y_dim(E);
    z_dim = dim.get_z_dim(E), stagewise_kkt_dim = dim.get_stagewise_kkt_dim(E);
    int xo = 0, yo = 0, zo = 0;
    for (int i = 0; i < N; ++i) {
      x_state_offsets[i] = xo;
      if (i < E) {
        xo += dim.get_state_dim(i);
        x_control_offsets[i] = xo;
        xo += dim.get_control_dim(i);
      }
    }
    for (int i = 0; i < N; ++i) {
      y_dyn_offsets[i] = yo, yo += dim.get_state_dim(i);
      y_node_c_offsets[i] = yo, yo += dim.get_node_c_dim(i);
    }
    for (int e = 0; e < E; ++e)
      y_edge_c_offsets[e] = yo, yo += dim.get_edge_c_dim(e);
    for (int i = 0; i < N; ++i)
      z_node_offsets[i] = zo, zo += dim.get_node_g_dim(i);
    for (int e = 0; e < E; ++e)
      z_edge_offsets[e] = zo, zo += dim.get_edge_g_dim(e);
  }
  void free(const Topology &topology) {
    model_callback_output.free(topology);
    for (int *t : {x_state_offsets, x_control_offsets, y_dyn_offsets, y_node_c_offsets, y_edge_c_offsets,
                   z_node_offsets, z_edge_offsets})
      delete[] t;
  }
};

class CallbackProvider {
public:
  CallbackProvider(const Input &input, Workspace &workspace) : input_(input), workspace_(workspace) {
    const auto &d = input.dimensions;
    const auto &t = input.topology;
    check(sip_kkt_plan_create(1, t.num_edges, t.root, t.edge_parents, t.edge_children, d.state_dims,
                              d.control_dims, d.node_c_dims, d.node_g_dims, d.edge_c_dims, d.edge_g_dims, 0,
                              &plan_) == SIP_LQR_OK && plan_ != nullptr,
          "sip_kkt_plan_create");
    input_is_valid_ = sip_kkt_input_status(plan_) == SIP_KKT_SUCCESS; // helpers.cpp:24-26
    if (!input_is_valid_)
      return;
    p_ = d.theta_dim;
    if (p_ > 0)
      check(sip_kkt_plan_set_theta(plan_, p_) == SIP_LQR_OK, "sip_kkt_plan_set_theta");
    sx_ = (int)sip_kkt_len(plan_, SIP_KKT_LEN_X), y_ = (int)sip_kkt_len(plan_, SIP_KKT_LEN_Y);
    z_ = (int)sip_kkt_len(plan_, SIP_KKT_LEN_Z);
    full_ = sx_ + p_ + y_ + z_;
    model_.assign(sip_kkt_len(plan_, SIP_KKT_LEN_MODEL) + 1, 0.0);
    theta_.assign(sip_kkt_theta_len(plan_) + 1, 0.0);
    d_model_ = dev(model_.size()), d_theta_ = dev(theta_.size());
    d_w_ = dev(z_), d_r1_ = dev(sx_ + p_), d_r2_ = dev(y_), d_r3_ = dev(z_);
    d_a_ = dev(full_), d_b_ = dev(full_);
    check(hipMalloc(&d_work_, sip_kkt_work_bytes(plan_) + 16) == hipSuccess, "hipMalloc");
    check(hipMalloc(&d_twork_, sip_kkt_theta_work_bytes(plan_) + 16) == hipSuccess, "hipMalloc");
    check(hipMalloc((void **)&d_status_, sizeof(int32_t)) == hipSuccess, "hipMalloc");
  }
  CallbackProvider(const CallbackProvider &) = delete;
  CallbackProvider &operator=(const CallbackProvider &) = delete;
  ~CallbackProvider() {
    for (void *q : {(void *)d_model_, (void *)d_theta_, (void *)d_w_, (void *)d_r1_, (void *)d_r2_, (void *)d_r3_,
                    (void *)d_a_, (void *)d_b_, d_work_, d_twork_, (void *)d_status_})
      if (q)
        (void)hipFree(q);
    sip_kkt_plan_destroy(plan_);
  }

  // helpers.cpp:242-407
  bool factor(const double *w, const double *r1, const double *r2, const double *r3) {
    if (!input_is_valid_)
      return false;
    gather_model();
    up(d_model_, model_.data(), model_.size()), up(d_w_, w, z_), up(d_r1_, r1, sx_ + p_);
    up(d_r2_, r2, y_), up(d_r3_, r3, z_);
    int rc;
    if (p_ > 0) {
      up(d_theta_, theta_.data(), theta_.size());
      rc = sip_kkt_factor_theta(plan_, d_model_, d_theta_, d_w_, d_r1_, d_r2_, d_r3_, d_work_, d_twork_, d_status_,
                                nullptr);
    } else {
      rc = sip_kkt_factor(plan_, d_model_, d_w_, d_r1_, d_r2_, d_r3_, d_work_, d_status_, nullptr);
    }
    check(rc == SIP_LQR_OK, "sip_kkt_factor");
    int32_t st = -1;
    check(hipMemcpy(&st, d_status_, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess, "status copy");
    return st == SIP_KKT_SUCCESS;
  }

  // helpers.cpp:896-951; assumes a prior successful factor(), like the reference
  void solve(const double *b, double *sol) {
    up(d_a_, b, full_);
    const int rc = p_ > 0 ? sip_kkt_solve_theta(plan_, d_model_, d_theta_, d_a_, d_b_, d_work_, d_twork_, d_status_,
                                                nullptr)
                          : sip_kkt_solve(plan_, d_model_, d_a_, d_b_, d_work_, d_status_, nullptr);
    check(rc == SIP_LQR_OK, "sip_kkt_solve");
    down(sol, d_b_, full_);
  }

  // helpers.cpp:953-976 (x_x has x_dim entries, theta included)
  void add_Kx_to_y(const double *w, const double *r1, const double *r2, const double *r3, const double *x_x,
                   const double *x_y, const double *x_z, double *y_x, double *y_y, double *y_z) {
    gather_model();
    up(d_model_, model_.data(), model_.size()), up(d_w_, w, z_), up(d_r1_, r1, sx_ + p_);
    up(d_r2_, r2, y_), up(d_r3_, r3, z_);
    std::vector<double> x(full_ + 1), y(full_ + 1);
    std::copy(x_x, x_x + sx_ + p_, x.begin());
    std::copy(x_y, x_y + y_, x.begin() + sx_ + p_);
    std::copy(x_z, x_z + z_, x.begin() + sx_ + p_ + y_);
    std::copy(y_x, y_x + sx_ + p_, y.begin());
    std::copy(y_y, y_y + y_, y.begin() + sx_ + p_);
    std::copy(y_z, y_z + z_, y.begin() + sx_ + p_ + y_);
    up(d_a_, x.data(), full_), up(d_b_, y.data(), full_);
    int rc;
    if (p_ > 0) {
      up(d_theta_, theta_.data(), theta_.size());
      rc = sip_kkt_add_Kx_to_y_theta(plan_, d_model_, d_theta_, d_w_, d_r1_, d_r2_, d_r3_, d_a_, d_b_, nullptr);
    } else {
      rc = sip_kkt_add_Kx_to_y(plan_, d_model_, d_w_, d_r1_, d_r2_, d_r3_, d_a_, d_b_, nullptr);
    }
    check(rc == SIP_LQR_OK, "sip_kkt_add_Kx_to_y");
    down(y.data(), d_b_, full_);
    std::copy(y.begin(), y.begin() + sx_ + p_, y_x);
    std::copy(y.begin() + sx_ + p_, y.begin() + sx_ + p_ + y_, y_y);
    std::copy(y.begin() + sx_ + p_ + y_, y.begin() + full_, y_z);
  }

  // helpers.hpp:20-24, bodies helpers.cpp:978-1368: the block operators SIP is handed one by one
  // (sip_optimal_control.cpp:147-190).  x-space vectors have x_dim entries (theta included),
  // y-space y_dim, z-space z_dim; every one accumulates into y.
  void add_Hx_to_y(const double *x, double *y) { block_op(0, x, sx_ + p_, y, sx_ + p_); }
  void add_Cx_to_y(const double *x, double *y) { block_op(1, x, sx_ + p_, y, y_); }
  void add_CTx_to_y(const double *x, double *y) { block_op(2, x, y_, y, sx_ + p_); }
  void add_Gx_to_y(const double *x, double *y) { block_op(3, x, sx_ + p_, y, z_); }
  void add_GTx_to_y(const double *x, double *y) { block_op(4, x, z_, y, sx_ + p_); }

private:
  void block_op(int which, const double *x, int x_len, double *y, int y_len) {
    typedef int (*plain_fn)(const sip_kkt_plan *, const double *, const double *, double *, void *);
    typedef int (*theta_fn)(const sip_kkt_plan *, const double *, const double *, const double *, double *, void *);
    static const plain_fn plain[5] = {sip_kkt_add_Hx_to_y, sip_kkt_add_Cx_to_y, sip_kkt_add_CTx_to_y,
                                      sip_kkt_add_Gx_to_y, sip_kkt_add_GTx_to_y};
    static const theta_fn with_theta[5] = {sip_kkt_add_Hx_to_y_theta, sip_kkt_add_Cx_to_y_theta,
                                           sip_kkt_add_CTx_to_y_theta, sip_kkt_add_Gx_to_y_theta,
                                           sip_kkt_add_GTx_to_y_theta};
    if (!input_is_valid_ || x_len == 0 || y_len == 0)
      return;
    gather_model(); // the callback outputs may have changed since the last call (helpers.cpp reads them live)
    up(d_model_, model_.data(), model_.size());
    up(d_a_, x, x_len), up(d_b_, y, y_len);
    int rc;
    if (p_ > 0) {
      up(d_theta_, theta_.data(), theta_.size());
      rc = with_theta[which](plan_, d_model_, d_theta_, d_a_, d_b_, nullptr);
    } else {
      rc = plain[which](plan_, d_model_, d_a_, d_b_, nullptr);
    }
    check(rc == SIP_LQR_OK, "sip_kkt_add_*x_to_y");
    down(y, d_b_, y_len);
  }
  static void check(bool ok, const char *what) {
    if (!ok) {
      std::fprintf(stderr, "sip_optimal_control_amd: %s failed (this adapter needs libsip_lqr_amd.so and a HIP device)\n",
                   what);
      std::abort();
    }
  }
  static double *dev(size_t count) {
    void *q = nullptr;
    check(hipMalloc(&q, (count > 0 ? count : 1) * sizeof(double)) == hipSuccess, "hipMalloc");
    return (double *)q;
  }
  static void up(double *dst, const double *src, size_t count) {
    if (count > 0)
      check(hipMemcpy(dst, src, count * sizeof(double), hipMemcpyHostToDevice) == hipSuccess, "H2D copy");
  }
  static void down(double *dst, const double *src, size_t count) {
    if (count > 0)
      check(hipMemcpy(dst, src, count * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess, "D2H copy");
  }
  void put(std::vector<double> &arena, size_t off, const double *src, int count) {
    if (count > 0)
      std::copy(src, src + count, arena.begin() + off);
  }
  // pointer-per-field callback outputs -> the flat arenas of sip_kkt_amd.h
  void gather_model() {
    const auto &d = input_.dimensions;
    const auto &t = input_.topology;
    const auto &mco = workspace_.model_callback_output;
    for (int i = 0; i < t.num_nodes(); ++i) {
      const int n = d.get_state_dim(i), c = d.get_node_c_dim(i), g = d.get_node_g_dim(i);
      const auto &o = mco.nodes[i];
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_NODE_D2L_DX2, i), o.d2L_dx2, n * n);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_NODE_DC_DX, i), o.dc_dx, c * n);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_NODE_DG_DX, i), o.dg_dx, g * n);
      if (p_ > 0) {
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_NODE_D2L_DXDTHETA, i), o.d2L_dxdtheta, n * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_NODE_DC_DTHETA, i), o.dc_dtheta, c * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_NODE_DG_DTHETA, i), o.dg_dtheta, g * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_NODE_D2L_DTHETA2, i), o.d2L_dtheta2, p_ * p_);
      }
    }
    for (int e = 0; e < t.num_edges; ++e) {
      const int n = d.get_state_dim(t.edge_parents[e]), nc = d.get_state_dim(t.edge_children[e]);
      const int m = d.get_control_dim(e), c = d.get_edge_c_dim(e), g = d.get_edge_g_dim(e);
      const auto &o = mco.edges[e];
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_D2L_DX2, e), o.d2L_dx2, n * n);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_D2L_DXDU, e), o.d2L_dxdu, n * m);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_D2L_DU2, e), o.d2L_du2, m * m);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_DDYN_DX, e), o.ddyn_dx, nc * n);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_DDYN_DU, e), o.ddyn_du, nc * m);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_DC_DX, e), o.dc_dx, c * n);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_DC_DU, e), o.dc_du, c * m);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_DG_DX, e), o.dg_dx, g * n);
      put(model_, sip_kkt_model_offset(plan_, SIP_KKT_EDGE_DG_DU, e), o.dg_du, g * m);
      if (p_ > 0) {
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_EDGE_D2L_DXDTHETA, e), o.d2L_dxdtheta, n * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_EDGE_D2L_DUDTHETA, e), o.d2L_dudtheta, m * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_EDGE_DDYN_DTHETA, e), o.ddyn_dtheta, nc * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_EDGE_DC_DTHETA, e), o.dc_dtheta, c * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_EDGE_DG_DTHETA, e), o.dg_dtheta, g * p_);
        put(theta_, sip_kkt_theta_offset(plan_, SIP_KKT_TH_EDGE_D2L_DTHETA2, e), o.d2L_dtheta2, p_ * p_);
      }
    }
  }

  const Input &input_;
  Workspace &workspace_;
  sip_kkt_plan *plan_ = nullptr;
  bool input_is_valid_ = false;
  int p_ = 0, sx_ = 0, y_ = 0, z_ = 0, full_ = 0;
  std::vector<double> model_, theta_;
  double *d_model_ = nullptr, *d_theta_ = nullptr, *d_w_ = nullptr, *d_r1_ = nullptr, *d_r2_ = nullptr,
         *d_r3_ = nullptr, *d_a_ = nullptr, *d_b_ = nullptr;
  void *d_work_ = nullptr, *d_twork_ = nullptr;
  int32_t *d_status_ = nullptr;
};

} // namespace sip::optimal_control
