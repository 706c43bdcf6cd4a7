// test_callback_provider.cpp -- the reference's tests of the Newton-KKT callbacks
// (tests/variable_dimensions_test.cpp:71-181, 265-363), re-expressed against the drop-in
// `sip::optimal_control::CallbackProvider` of
// include/sip_optimal_control_amd/callback_provider_dropin.hpp, whose factor / solve /
// add_Kx_to_y run on the GPU.  No gtest here: a tiny check harness.
//
// Build (also done by __graft_entry__.build()): like tests/cpp/test_dropin.cpp.
#include "sip_optimal_control_amd/callback_provider_dropin.hpp"

#include <array>
#include <cmath>
#include <cstdio>
#include <functional>
#include <vector>

using namespace sip::optimal_control;

static int g_failures = 0, g_checks = 0;
#define CHECK(cond)                                                                       \
  do {                                                                                    \
    ++g_checks;                                                                           \
    if (!(cond)) {                                                                        \
      ++g_failures;                                                                       \
      std::printf("  CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);               \
    }                                                                                     \
  } while (0)

static void fill_sequence(double *data, int size, double scale) { // :46-50
  for (int i = 0; i < size; ++i)
    data[i] = scale * static_cast<double>(i + 1);
}
static void fill_spd(double *data, int size, double diagonal) { // :52-56
  for (int i = 0; i < size * size; ++i)
    data[i] = 0.0;
  for (int i = 0; i < size; ++i)
    data[i + size * i] = diagonal;
}

// initialize_model, :71-133
static void initialize_model(const Input &input, Workspace &workspace, double theta_diagonal = 0.0) {
  const int p = input.dimensions.theta_dim;
  auto &mco = workspace.model_callback_output;
  for (int node = 0; node < input.topology.num_nodes(); ++node) {
    const int n = input.dimensions.get_state_dim(node), c = input.dimensions.get_node_c_dim(node);
    const int g = input.dimensions.get_node_g_dim(node);
    auto &o = mco.nodes[node];
    fill_sequence(o.dc_dx, c * n, 0.013 * (node + 1));
    fill_sequence(o.dc_dtheta, c * p, 0.001 * (node + 1));
    fill_sequence(o.dg_dx, g * n, -0.011 * (node + 1));
    fill_sequence(o.dg_dtheta, g * p, -0.0007 * (node + 1));
    fill_spd(o.d2L_dx2, n, 2.5 + 0.2 * node);
    fill_sequence(o.d2L_dxdtheta, n * p, 0.0005 * (node + 1));
    fill_spd(o.d2L_dtheta2, p, theta_diagonal);
  }
  for (int edge = 0; edge < input.topology.num_edges; ++edge) {
    const int n_parent = input.dimensions.get_state_dim(input.topology.edge_parents[edge]);
    const int n_child = input.dimensions.get_state_dim(input.topology.edge_children[edge]);
    const int m = input.dimensions.get_control_dim(edge), c = input.dimensions.get_edge_c_dim(edge);
    const int g = input.dimensions.get_edge_g_dim(edge);
    auto &o = mco.edges[edge];
    fill_sequence(o.ddyn_dx, n_child * n_parent, 0.025 + 0.004 * edge);
    fill_sequence(o.ddyn_du, n_child * m, -0.031 - 0.003 * edge);
    fill_sequence(o.ddyn_dtheta, n_child * p, 0.0009 * (edge + 1));
    fill_sequence(o.dc_dx, c * n_parent, 0.017 * (edge + 1));
    fill_sequence(o.dc_du, c * m, 0.019 * (edge + 1));
    fill_sequence(o.dc_dtheta, c * p, 0.0008 * (edge + 1));
    fill_sequence(o.dg_dx, g * n_parent, -0.014 * (edge + 1));
    fill_sequence(o.dg_du, g * m, 0.016 * (edge + 1));
    fill_sequence(o.dg_dtheta, g * p, -0.0006 * (edge + 1));
    fill_spd(o.d2L_dx2, n_parent, 0.3 + 0.05 * edge);
    fill_sequence(o.d2L_dxdu, n_parent * m, 0.009 * (edge + 1));
    fill_spd(o.d2L_du2, m, 3.0 + 0.2 * edge);
    fill_sequence(o.d2L_dxdtheta, n_parent * p, 0.0004 * (edge + 1));
    fill_sequence(o.d2L_dudtheta, m * p, -0.0003 * (edge + 1));
    fill_spd(o.d2L_dtheta2, p, theta_diagonal);
  }
}

// expect_kkt_solve, :135-181
static void expect_kkt_solve(const Input &input, Workspace &workspace, double tolerance = 1e-9) {
  CallbackProvider callback_provider(input, workspace);
  const int E = input.topology.num_edges;
  const int x_dim = input.dimensions.get_x_dim(E), y_dim = input.dimensions.get_y_dim(E);
  const int z_dim = input.dimensions.get_z_dim(E), kkt_dim = x_dim + y_dim + z_dim;
  std::vector<double> w(z_dim + 1, 1.3), r2(y_dim + 1, 0.9), r3(z_dim + 1, 0.4), r1(x_dim + 1);
  fill_sequence(r1.data(), x_dim, 0.03);
  for (double &v : r1)
    v += 0.2;
  CHECK(callback_provider.factor(w.data(), r1.data(), r2.data(), r3.data()));
  std::vector<double> rhs(kkt_dim + 1), solution(kkt_dim + 1, 0.0);
  fill_sequence(rhs.data(), kkt_dim, 0.01);
  callback_provider.solve(rhs.data(), solution.data());
  std::vector<double> px(x_dim + 1, 0.0), py(y_dim + 1, 0.0), pz(z_dim + 1, 0.0);
  callback_provider.add_Kx_to_y(w.data(), r1.data(), r2.data(), r3.data(), solution.data(), solution.data() + x_dim,
                                solution.data() + x_dim + y_dim, px.data(), py.data(), pz.data());
  double sq = 0.0;
  for (int i = 0; i < x_dim; ++i)
    sq += (px[i] - rhs[i]) * (px[i] - rhs[i]);
  for (int i = 0; i < y_dim; ++i)
    sq += (py[i] - rhs[x_dim + i]) * (py[i] - rhs[x_dim + i]);
  for (int i = 0; i < z_dim; ++i)
    sq += (pz[i] - rhs[x_dim + y_dim + i]) * (pz[i] - rhs[x_dim + y_dim + i]);
  std::printf("    kkt_dim %d  ||K sol - rhs|| = %.3e (tolerance %.0e)\n", kkt_dim, std::sqrt(sq), tolerance);
  CHECK(std::sqrt(sq) < tolerance);
  // The same product from the five block operators, composed exactly as CallbackProvider::add_Kx_to_y
  // composes them (helpers.cpp:953-976): what SIP does with the callbacks it is handed one by one
  // (sip_optimal_control.cpp:147-190).
  std::vector<double> qx(x_dim + 1, 0.0), qy(y_dim + 1, 0.0), qz(z_dim + 1, 0.0);
  const double *x_x = solution.data(), *x_y = solution.data() + x_dim, *x_z = solution.data() + x_dim + y_dim;
  callback_provider.add_Hx_to_y(x_x, qx.data());
  callback_provider.add_Cx_to_y(x_x, qy.data());
  callback_provider.add_CTx_to_y(x_y, qx.data());
  callback_provider.add_Gx_to_y(x_x, qz.data());
  callback_provider.add_GTx_to_y(x_z, qx.data());
  for (int i = 0; i < x_dim; ++i)
    qx[i] += r1[i] * x_x[i];
  for (int i = 0; i < y_dim; ++i)
    qy[i] -= r2[i] * x_y[i];
  for (int i = 0; i < z_dim; ++i)
    qz[i] -= (w[i] + r3[i]) * x_z[i];
  double dq = 0.0;
  for (int i = 0; i < x_dim; ++i)
    dq = std::fmax(dq, std::fabs(qx[i] - px[i]));
  for (int i = 0; i < y_dim; ++i)
    dq = std::fmax(dq, std::fabs(qy[i] - py[i]));
  for (int i = 0; i < z_dim; ++i)
    dq = std::fmax(dq, std::fabs(qz[i] - pz[i]));
  std::printf("    five block operators + diagonal vs add_Kx_to_y: max |diff| = %.3e\n", dq);
  CHECK(dq < 1e-12);
}

struct Case {
  std::array<int, 3> state_dims;
  std::array<int, 2> control_dims;
  std::array<int, 3> node_c, node_g;
  std::array<int, 2> edge_c, edge_g;
  std::array<int, 2> parent, child;
  int theta_dim;
  double theta_diagonal, tolerance;
};

static void run_case(const Case &k) {
  Input input{{k.theta_dim, k.state_dims.data(), k.control_dims.data(), k.node_c.data(), k.node_g.data(),
               k.edge_c.data(), k.edge_g.data()},
              {2, 0, k.parent.data(), k.child.data()}};
  Workspace workspace;
  workspace.reserve(input.dimensions, input.topology);
  initialize_model(input, workspace, k.theta_diagonal);
  expect_kkt_solve(input, workspace, k.tolerance);
  workspace.free(input.topology);
}

static void run(const char *name, const std::function<void()> &fn) {
  const int before = g_failures;
  std::printf("[ RUN  ] %s\n", name);
  fn();
  std::printf("%s %s\n", g_failures == before ? "[  OK  ]" : "[ FAIL ]", name);
}

int main() {
  // CallbackProvider.SolvesChainWithNodeAndEdgeConstraints, :265-288
  run("CallbackProvider.SolvesChainWithNodeAndEdgeConstraints", [] {
    run_case({{2, 1, 3}, {1, 2}, {1, 0, 2}, {0, 2, 1}, {1, 2}, {2, 1}, {0, 1}, {1, 2}, 0, 0.0, 1e-9});
  });
  // CallbackProvider.SolvesIndependentConstraintsOnSiblingEdges, :290-313
  run("CallbackProvider.SolvesIndependentConstraintsOnSiblingEdges", [] {
    run_case({{2, 1, 3}, {1, 2}, {1, 0, 1}, {1, 1, 0}, {2, 1}, {1, 2}, {0, 0}, {1, 2}, 0, 0.0, 1e-9});
  });
  // CallbackProvider.SolvesBranchedSystemWithZeroDimensionalRoot, :315-336
  run("CallbackProvider.SolvesBranchedSystemWithZeroDimensionalRoot", [] {
    run_case({{0, 1, 3}, {1, 2}, {0, 0, 0}, {0, 0, 0}, {0, 0}, {0, 0}, {0, 0}, {1, 2}, 0, 0.0, 1e-9});
  });
  // CallbackProvider.SolvesBranchedSystemWithSchurVariables, :338-363
  run("CallbackProvider.SolvesBranchedSystemWithSchurVariables", [] {
    run_case({{2, 1, 3}, {1, 2}, {1, 0, 1}, {0, 1, 1}, {1, 2}, {2, 1}, {0, 0}, {1, 2}, 2, 6.0, 1e-8});
  });
  // factor() is false on an invalid input (InputValidation, :183-228; helpers.cpp:244-246) and
  // on a non-positive regularization (helpers.cpp:256-297)
  run("CallbackProvider.FactorFalsePaths", [] {
    const std::array<int, 3> sd = {2, 1, 3}, zero3 = {0, 0, 0};
    const std::array<int, 2> cd = {1, 2}, neg = {-1, 1}, zero2 = {0, 0}, pa = {0, 1}, ch = {2, 2}, cpa = {0, 1},
                             cch = {1, 2};
    {
      Input bad{{0, sd.data(), cd.data(), zero3.data(), zero3.data(), neg.data(), zero2.data()},
                {2, 0, cpa.data(), cch.data()}};
      Workspace ws; // never dereferenced: the input does not validate
      CallbackProvider cp(bad, ws);
      CHECK(!cp.factor(nullptr, nullptr, nullptr, nullptr));
    }
    {
      Input dag{{0, sd.data(), cd.data(), zero3.data(), zero3.data(), zero2.data(), zero2.data()},
                {2, 0, pa.data(), ch.data()}}; // node 2 has two parents (:210-214)
      Workspace ws;
      CallbackProvider cp(dag, ws);
      CHECK(!cp.factor(nullptr, nullptr, nullptr, nullptr));
    }
    {
      Input ok{{0, sd.data(), cd.data(), zero3.data(), zero3.data(), zero2.data(), zero2.data()},
               {2, 0, cpa.data(), cch.data()}};
      Workspace ws;
      ws.reserve(ok.dimensions, ok.topology);
      initialize_model(ok, ws);
      CallbackProvider cp(ok, ws);
      std::vector<double> w(1, 1.0), r3(1, 1.0), r1(ws.x_dim, 0.5), r2(ws.y_dim, 0.9);
      CHECK(cp.factor(w.data(), r1.data(), r2.data(), r3.data()));
      r2[ws.y_dyn_offsets[1]] = 0.0;
      CHECK(!cp.factor(w.data(), r1.data(), r2.data(), r3.data()));
      ws.free(ok.topology);
    }
  });
  std::printf("%d checks, %d failures\n", g_checks, g_failures);
  return g_failures == 0 ? 0 : 1;
}
