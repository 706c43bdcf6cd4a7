// test_dropin.cpp -- the reference's hot-path tests (tests/lqr_test.cpp),
// re-expressed against the drop-in `sip::optimal_control::LQR` of
// include/sip_optimal_control_amd/lqr_dropin.hpp, whose factor/solve run on
// the GPU.  No gtest / Eigen here: a tiny check harness, column-major
// std::vector blocks, a Gaussian-elimination dense KKT solve.
//
// Build (also done by __graft_entry__.build()):
//   g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ tests/cpp/test_dropin.cpp -I include -I /opt/rocm/include \
//     -L sip_optimal_control_amd/lib -lsip_lqr_amd -L /opt/rocm/lib -lamdhip64 \
//     -Wl,-rpath,'$ORIGIN/../../sip_optimal_control_amd/lib' -Wl,-rpath,/opt/rocm/lib -o tests/cpp/test_dropin
#include "sip_optimal_control_amd/lqr_dropin.hpp"

#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

using namespace sip::optimal_control;
using Status = LQR::FactorStatus;

static int g_failures = 0, g_checks = 0;
#define CHECK(cond)                                                                       \
  do {                                                                                    \
    ++g_checks;                                                                           \
    if (!(cond)) {                                                                        \
      ++g_failures;                                                                       \
      std::printf("  CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);               \
    }                                                                                     \
  } while (0)

// Column-major dense block.
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int r, int c, double v = 0.0) : rows(r), cols(c), a((size_t)r * c, v) {}
  double &operator()(int i, int j) { return a[i + (size_t)j * rows]; }
  double operator()(int i, int j) const { return a[i + (size_t)j * rows]; }
  static Mat identity(int n, double s = 1.0) {
    Mat m(n, n);
    for (int i = 0; i < n; ++i)
      m(i, i) = s;
    return m;
  }
  // rows given row-major, like an Eigen comma initialiser
  static Mat from_rows(int r, int c, std::initializer_list<double> v) {
    Mat m(r, c);
    auto it = v.begin();
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < c; ++j)
        m(i, j) = *it++;
    return m;
  }
};
using Vec = std::vector<double>;

static Vec linspaced(int n, double lo, double hi) { // Eigen::VectorXd::LinSpaced
  Vec v(n);
  if (n == 1) {
    v[0] = hi;
    return v;
  }
  for (int i = 0; i < n; ++i)
    v[i] = lo + (hi - lo) * i / (n - 1);
  return v;
}

struct Problem {
  std::vector<int> parents, children, state_dims, control_dims;
  int root = 0;
  std::vector<Mat> Q, M, R, A, B;
  std::vector<Vec> q, r, c, delta;
  std::vector<double *> Qp, Mp, Rp, Ap, Bp, qp, rp, cp, dp;
  Dimensions dims;
  Topology topo;

  int E() const { return (int)control_dims.size(); }
  int N() const { return E() + 1; }
  LQR::Input input() {
    auto tab_m = [](std::vector<Mat> &v, std::vector<double *> &p) {
      p.resize(v.size());
      for (size_t i = 0; i < v.size(); ++i)
        p[i] = v[i].a.data();
    };
    auto tab_v = [](std::vector<Vec> &v, std::vector<double *> &p) {
      p.resize(v.size());
      for (size_t i = 0; i < v.size(); ++i)
        p[i] = v[i].data();
    };
    tab_m(Q, Qp), tab_m(M, Mp), tab_m(R, Rp), tab_m(A, Ap), tab_m(B, Bp);
    tab_v(q, qp), tab_v(r, rp), tab_v(c, cp), tab_v(delta, dp);
    dims = Dimensions{0, state_dims.data(), control_dims.data(), nullptr, nullptr};
    topo = Topology{E(), root, parents.data(), children.data()};
    return LQR::Input{Qp.data(), Mp.data(), Rp.data(), qp.data(), rp.data(),
                      Ap.data(), Bp.data(), cp.data(), dp.data(), dims,        topo};
  }
};

struct Solution {
  std::vector<Vec> x, u, y;
  std::vector<double *> xp, up, yp;
  explicit Solution(const Problem &p) {
    for (int n : p.state_dims)
      x.emplace_back(n, 0.0), y.emplace_back(n, 0.0);
    for (int m : p.control_dims)
      u.emplace_back(m, 0.0);
    for (auto &v : x) xp.push_back(v.data());
    for (auto &v : u) up.push_back(v.data());
    for (auto &v : y) yp.push_back(v.data());
  }
  LQR::Output output() { return LQR::Output{xp.data(), up.data(), yp.data()}; }
};

// LQRProblem(n, m, T) defaults, lqr_test.cpp:45-75.
static Problem default_chain(int n, int m, int T) {
  Problem p;
  for (int e = 0; e < T; ++e)
    p.parents.push_back(e), p.children.push_back(e + 1), p.control_dims.push_back(m);
  p.state_dims.assign(T + 1, n);
  for (int i = 0; i <= T; ++i) {
    p.Q.push_back(Mat::identity(n));
    p.q.emplace_back(n, 0.0), p.c.emplace_back(n, 0.0), p.delta.emplace_back(n, 1.0);
  }
  for (int i = 0; i < T; ++i) {
    p.M.emplace_back(n, m, 0.0), p.R.push_back(Mat::identity(m)), p.A.push_back(Mat::identity(n));
    p.B.emplace_back(n, m, 1.0), p.r.emplace_back(m, 0.0);
  }
  return p;
}

// lqr_test.cpp:229-247
static Problem nonuniform_diagonal_delta() {
  Problem p = default_chain(3, 2, 3);
  for (int i = 0; i < 3; ++i) {
    p.A[i] = Mat::from_rows(3, 3, {1.0 + 0.02 * i, 0.03, -0.01, -0.02, 0.95 + 0.01 * i, 0.04, 0.01, -0.03,
                                   1.02 - 0.01 * i});
    p.B[i] = Mat::from_rows(3, 2, {0.2, -0.1, 0.05, 0.15, -0.1, 0.08});
    p.Q[i] = Mat(3, 3);
    p.Q[i](0, 0) = 1.0 + 0.1 * i, p.Q[i](1, 1) = 1.4 + 0.05 * i, p.Q[i](2, 2) = 1.8 + 0.03 * i;
    p.R[i] = Mat(2, 2);
    p.R[i](0, 0) = 1.2 + 0.1 * i, p.R[i](1, 1) = 1.6 + 0.07 * i;
    p.q[i] = {0.2 + 0.01 * i, -0.1 + 0.02 * i, 0.05 - 0.03 * i};
    p.r[i] = {-0.2 + 0.03 * i, 0.1 - 0.01 * i};
    p.c[i] = {0.03 + 0.01 * i, -0.04 + 0.02 * i, 0.02 - 0.01 * i};
    p.delta[i] = {0.03 + 0.01 * i, 0.11 + 0.02 * i, 0.19 + 0.03 * i};
  }
  p.Q[3] = Mat(3, 3);
  p.Q[3](0, 0) = 1.3, p.Q[3](1, 1) = 1.7, p.Q[3](2, 2) = 2.1;
  p.q[3] = {0.06, -0.08, 0.12}, p.c[3] = {-0.02, 0.05, -0.01}, p.delta[3] = {0.07, 0.17, 0.29};
  return p;
}

// lqr_test.cpp:300-335
static Problem branch_tree() {
  Problem p;
  p.parents = {0, 0}, p.children = {1, 2}, p.state_dims = {2, 2, 2}, p.control_dims = {1, 1};
  p.Q = {Mat::from_rows(2, 2, {2.0, 0.1, 0.1, 1.5}), Mat::from_rows(2, 2, {1.3, 0.2, 0.2, 1.7}),
         Mat::from_rows(2, 2, {1.8, -0.1, -0.1, 1.4})};
  p.M = {Mat::from_rows(2, 1, {0.2, -0.1}), Mat::from_rows(2, 1, {-0.15, 0.05})};
  p.R = {Mat::from_rows(1, 1, {1.6}), Mat::from_rows(1, 1, {1.9})};
  p.A = {Mat::from_rows(2, 2, {1.0, 0.2, 0.0, 0.9}), Mat::from_rows(2, 2, {0.8, -0.1, 0.3, 1.1})};
  p.B = {Mat::from_rows(2, 1, {0.4, 0.2}), Mat::from_rows(2, 1, {-0.1, 0.5})};
  p.q = {{0.3, -0.2}, {-0.1, 0.4}, {0.2, 0.1}};
  p.r = {{-0.3}, {0.25}};
  p.c = {{0.1, -0.2}, {-0.05, 0.1}, {0.2, 0.15}};
  p.delta = {{0.7, 0.9}, {0.8, 1.1}, {1.0, 0.6}};
  return p;
}

// lqr_test.cpp:494-532
static Problem variable_dimension_branch() {
  Problem p;
  p.parents = {0, 0}, p.children = {1, 2}, p.state_dims = {2, 1, 3}, p.control_dims = {2, 1};
  p.Q = {Mat::from_rows(2, 2, {2.0, 0.1, 0.1, 1.7}), Mat::from_rows(1, 1, {1.3}),
         Mat::from_rows(3, 3, {1.8, 0.1, -0.2, 0.1, 1.6, 0.05, -0.2, 0.05, 2.1})};
  p.M = {Mat::from_rows(2, 2, {0.1, -0.2, 0.05, 0.15}), Mat::from_rows(2, 1, {-0.1, 0.2})};
  p.R = {Mat::from_rows(2, 2, {1.8, 0.1, 0.1, 1.5}), Mat::from_rows(1, 1, {1.4})};
  p.A = {Mat::from_rows(1, 2, {0.8, -0.3}), Mat::from_rows(3, 2, {1.0, 0.2, -0.1, 0.7, 0.3, -0.4})};
  p.B = {Mat::from_rows(1, 2, {0.4, -0.2}), Mat::from_rows(3, 1, {0.2, -0.1, 0.5})};
  p.q = {{0.2, -0.15}, {-0.05}, {0.1, -0.2, 0.05}};
  p.r = {{-0.1, 0.25}, {-0.2}};
  p.c = {{0.05, -0.1}, {0.12}, {-0.02, 0.04, -0.08}};
  p.delta = {{0.8, 1.1}, {0.9}, {0.7, 1.0, 1.2}};
  return p;
}

// lqr_test.cpp:661-762
static Problem five_node_tree() {
  Problem p;
  p.parents = {0, 0, 1, 1}, p.children = {1, 2, 3, 4};
  p.state_dims = {3, 1, 2, 4, 2}, p.control_dims = {2, 1, 3, 1};
  for (int node = 0; node < 5; ++node) {
    const int n = p.state_dims[node];
    Mat Q = Mat::identity(n, 1.5 + 0.2 * node);
    for (int col = 0; col < n; ++col)
      for (int row = col + 1; row < n; ++row)
        Q(row, col) = Q(col, row) = 0.02 * (row + col + node + 1);
    p.Q.push_back(Q);
    p.q.push_back(linspaced(n, -0.15 + 0.03 * node, 0.12 + 0.02 * node));
    p.c.push_back(linspaced(n, 0.05 * node, 0.04 + 0.03 * node));
    p.delta.push_back(linspaced(n, 0.7 + 0.05 * node, 1.0 + 0.04 * node));
  }
  for (int e = 0; e < 4; ++e) {
    const int np = p.state_dims[p.parents[e]], nc = p.state_dims[p.children[e]], m = p.control_dims[e];
    Mat M(np, m), A(nc, np), B(nc, m);
    for (int col = 0; col < m; ++col)
      for (int row = 0; row < np; ++row)
        M(row, col) = 0.015 * ((e + 1) * (row + 1) - col);
    for (int col = 0; col < np; ++col)
      for (int row = 0; row < nc; ++row)
        A(row, col) = 0.08 * (row + 1) / (double)(e + col + 2);
    for (int col = 0; col < m; ++col)
      for (int row = 0; row < nc; ++row)
        B(row, col) = -0.06 * (col + 1) / (double)(e + row + 2);
    Mat R = Mat::identity(m, 1.8 + 0.1 * e);
    for (int col = 0; col < m; ++col)
      for (int row = col + 1; row < m; ++row)
        R(row, col) = R(col, row) = 0.03 * (row + col + 1);
    p.M.push_back(M), p.A.push_back(A), p.B.push_back(B), p.R.push_back(R);
    p.r.push_back(linspaced(m, -0.2 + 0.04 * e, 0.1 + 0.03 * e));
  }
  return p;
}

// KKT residual of lqr_test.cpp:152-186 / 371-409 / 600-639 (Q, R taken as
// selfadjointView<Lower>).
static double kkt_residual(const Problem &p, const Solution &s) {
  double sq = 0.0;
  auto symmv = [](const Mat &S, const Vec &v, Vec &out) {
    for (int i = 0; i < S.rows; ++i)
      for (int j = 0; j < S.cols; ++j)
        out[i] += (i >= j ? S(i, j) : S(j, i)) * v[j];
  };
  for (int node = 0; node < p.N(); ++node) {
    const int n = p.state_dims[node];
    Vec res(n, 0.0);
    symmv(p.Q[node], s.x[node], res);
    for (int i = 0; i < n; ++i)
      res[i] += p.q[node][i] - s.y[node][i];
    for (int e = 0; e < p.E(); ++e)
      if (p.parents[e] == node) {
        const int ch = p.children[e];
        for (int i = 0; i < n; ++i) {
          for (int j = 0; j < p.control_dims[e]; ++j)
            res[i] += p.M[e](i, j) * s.u[e][j];
          for (int j = 0; j < p.state_dims[ch]; ++j)
            res[i] += p.A[e](j, i) * s.y[ch][j];
        }
      }
    for (double v : res) sq += v * v;
  }
  for (int e = 0; e < p.E(); ++e) {
    const int pa = p.parents[e], ch = p.children[e], m = p.control_dims[e];
    const int np = p.state_dims[pa], nc = p.state_dims[ch];
    Vec su(m, 0.0), dy(nc, 0.0);
    symmv(p.R[e], s.u[e], su);
    for (int i = 0; i < m; ++i) {
      su[i] += p.r[e][i];
      for (int j = 0; j < np; ++j) su[i] += p.M[e](j, i) * s.x[pa][j];
      for (int j = 0; j < nc; ++j) su[i] += p.B[e](j, i) * s.y[ch][j];
    }
    for (int i = 0; i < nc; ++i) {
      dy[i] = -s.x[ch][i] + p.c[ch][i] - p.delta[ch][i] * s.y[ch][i];
      for (int j = 0; j < np; ++j) dy[i] += p.A[e](i, j) * s.x[pa][j];
      for (int j = 0; j < m; ++j) dy[i] += p.B[e](i, j) * s.u[e][j];
    }
    for (double v : su) sq += v * v;
    for (double v : dy) sq += v * v;
  }
  for (int i = 0; i < p.state_dims[p.root]; ++i) {
    const double v = -s.x[p.root][i] - p.delta[p.root][i] * s.y[p.root][i] + p.c[p.root][i];
    sq += v * v;
  }
  return std::sqrt(sq);
}

// Dense KKT assembly of lqr_test.cpp:859-929, solved by Gaussian elimination
// with partial pivoting.  Returns [x nodes | u edges | y nodes].
static Vec dense_kkt(const Problem &p) {
  std::vector<int> xo(p.N()), uo(p.E()), yo(p.N());
  int total = 0;
  for (int i = 0; i < p.N(); ++i) xo[i] = total, total += p.state_dims[i];
  for (int e = 0; e < p.E(); ++e) uo[e] = total, total += p.control_dims[e];
  for (int i = 0; i < p.N(); ++i) yo[i] = total, total += p.state_dims[i];
  Mat K(total, total);
  Vec rhs(total, 0.0);
  int row = 0;
  for (int node = 0; node < p.N(); ++node) {
    const int n = p.state_dims[node];
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) K(row + i, xo[node] + j) += p.Q[node](i, j);
      K(row + i, yo[node] + i) -= 1.0;
      rhs[row + i] = -p.q[node][i];
    }
    for (int e = 0; e < p.E(); ++e)
      if (p.parents[e] == node) {
        const int ch = p.children[e];
        for (int i = 0; i < n; ++i) {
          for (int j = 0; j < p.control_dims[e]; ++j) K(row + i, uo[e] + j) += p.M[e](i, j);
          for (int j = 0; j < p.state_dims[ch]; ++j) K(row + i, yo[ch] + j) += p.A[e](j, i);
        }
      }
    row += n;
  }
  for (int e = 0; e < p.E(); ++e) {
    const int pa = p.parents[e], ch = p.children[e], m = p.control_dims[e];
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < p.state_dims[pa]; ++j) K(row + i, xo[pa] + j) += p.M[e](j, i);
      for (int j = 0; j < m; ++j) K(row + i, uo[e] + j) += p.R[e](i, j);
      for (int j = 0; j < p.state_dims[ch]; ++j) K(row + i, yo[ch] + j) += p.B[e](j, i);
      rhs[row + i] = -p.r[e][i];
    }
    row += m;
  }
  for (int i = 0; i < p.state_dims[p.root]; ++i) {
    K(row + i, xo[p.root] + i) -= 1.0;
    K(row + i, yo[p.root] + i) -= p.delta[p.root][i];
    rhs[row + i] = -p.c[p.root][i];
  }
  row += p.state_dims[p.root];
  for (int e = 0; e < p.E(); ++e) {
    const int pa = p.parents[e], ch = p.children[e], nc = p.state_dims[ch];
    for (int i = 0; i < nc; ++i) {
      for (int j = 0; j < p.state_dims[pa]; ++j) K(row + i, xo[pa] + j) += p.A[e](i, j);
      for (int j = 0; j < p.control_dims[e]; ++j) K(row + i, uo[e] + j) += p.B[e](i, j);
      K(row + i, xo[ch] + i) -= 1.0;
      K(row + i, yo[ch] + i) -= p.delta[ch][i];
      rhs[row + i] = -p.c[ch][i];
    }
    row += nc;
  }
  for (int k = 0; k < total; ++k) { // elimination
    int piv = k;
    for (int i = k + 1; i < total; ++i)
      if (std::fabs(K(i, k)) > std::fabs(K(piv, k))) piv = i;
    if (piv != k) {
      for (int j = 0; j < total; ++j) std::swap(K(k, j), K(piv, j));
      std::swap(rhs[k], rhs[piv]);
    }
    for (int i = k + 1; i < total; ++i) {
      const double f = K(i, k) / K(k, k);
      if (f == 0.0) continue;
      for (int j = k; j < total; ++j) K(i, j) -= f * K(k, j);
      rhs[i] -= f * rhs[k];
    }
  }
  Vec z(total);
  for (int i = total - 1; i >= 0; --i) {
    double s = rhs[i];
    for (int j = i + 1; j < total; ++j) s -= K(i, j) * z[j];
    z[i] = s / K(i, i);
  }
  return z;
}

static Status factor_status(Problem &p) { // lqr_test.cpp:142-150
  auto input = p.input();
  LQR::Workspace ws;
  ws.reserve(input.dimensions, input.topology);
  Status st;
  {
    auto lqr = LQR(input, ws);
    st = lqr.factor_with_status();
  }
  ws.free(p.E());
  return st;
}

static double solve_residual(Problem &p, bool uniform_reserve) {
  auto input = p.input();
  LQR::Workspace ws;
  if (uniform_reserve)
    ws.reserve(p.state_dims[0], p.control_dims[0], p.E());
  else
    ws.reserve(input.dimensions, input.topology);
  double res;
  {
    auto lqr = LQR(input, ws);
    CHECK(lqr.factor_with_status() == Status::SUCCESS);
    Solution s(p);
    auto out = s.output();
    lqr.solve(out);
    res = kkt_residual(p, s);
  }
  ws.free(p.E());
  return res;
}

int main() {
  struct Case { const char *name; std::function<void()> run; };
  std::vector<Case> cases = {
      {"LQRFactor.ReportsSuccess", [] {
         auto p = default_chain(2, 1, 2);
         CHECK(factor_status(p) == Status::SUCCESS);
       }},
      {"LQRFactor.BoolFactorWrapsStatusApi", [] {
         auto p = default_chain(2, 1, 2);
         auto input = p.input();
         LQR::Workspace ws;
         ws.reserve(2, 1, 2);
         {
           auto lqr = LQR(input, ws);
           CHECK(lqr.factor());
         }
         ws.free(2);
       }},
      {"LQRFactor.ReportsInvalidDelta", [] {
         auto p = default_chain(2, 1, 2);
         p.delta[2][0] = 0.0;
         CHECK(factor_status(p) == Status::INVALID_DELTA);
       }},
      {"LQRFactor.ReportsFFactorizationFailure", [] {
         auto p = default_chain(1, 1, 1);
         p.Q[1](0, 0) = -2.0, p.delta[1][0] = 1.0;
         CHECK(factor_status(p) == Status::F_FACTORIZATION_FAILURE);
       }},
      {"LQRFactor.ReportsGFactorizationFailure", [] {
         auto p = default_chain(1, 1, 1);
         p.Q[1](0, 0) = 0.0, p.R[0](0, 0) = -1.0;
         CHECK(factor_status(p) == Status::G_FACTORIZATION_FAILURE);
       }},
      {"LQRSolve.SolvesNonuniformDiagonalDeltaProblem", [] {
         auto p = nonuniform_diagonal_delta();
         CHECK(solve_residual(p, true) < 1e-12);
       }},
      {"LQRSolve.SolvesBranchingTreeProblem", [] {
         auto p = branch_tree();
         CHECK(solve_residual(p, true) < 1e-12);
       }},
      {"LQRTopology.ReusesCompiledTopologyAcrossFactorAndSolveCalls", [] {
         auto p = branch_tree();
         auto input = p.input();
         LQR::Workspace ws;
         ws.reserve(2, 1, 2);
         {
           auto lqr = LQR(input, ws);
           CHECK(lqr.factor_with_status() == Status::SUCCESS);
           CHECK(lqr.factor_with_status() == Status::SUCCESS);
           Solution s(p);
           auto out = s.output();
           lqr.solve(out);
           lqr.solve(out);
           CHECK(kkt_residual(p, s) < 1e-12);
         }
         ws.free(2);
       }},
      {"LQRFactor.RejectsInvalidTreeTopology", [] {
         auto p = branch_tree();
         p.children[1] = 1;
         CHECK(factor_status(p) == Status::INVALID_TOPOLOGY);
       }},
      {"LQRSolve.SolvesVariableDimensionBranchingTreeProblem", [] {
         auto p = variable_dimension_branch();
         CHECK(solve_residual(p, false) < 1e-12);
       }},
      {"LQRTopology.CompilesMultiChildPreorderAndPostorder", [] {
         auto p = five_node_tree();
         auto input = p.input();
         LQR::Workspace ws;
         ws.reserve(input.dimensions, input.topology);
         {
           auto lqr = LQR(input, ws);
           CHECK(lqr.factor_with_status() == Status::SUCCESS);
           CHECK(std::vector<int>(ws.child_offsets, ws.child_offsets + 6) == (std::vector<int>{0, 2, 4, 4, 4, 4}));
           CHECK(std::vector<int>(ws.child_edges, ws.child_edges + 4) == (std::vector<int>{0, 1, 2, 3}));
           CHECK(std::vector<int>(ws.preorder_nodes, ws.preorder_nodes + 5) == (std::vector<int>{0, 1, 3, 4, 2}));
           CHECK(std::vector<int>(ws.postorder_nodes, ws.postorder_nodes + 5) == (std::vector<int>{2, 4, 3, 1, 0}));
         }
         ws.free(4);
       }},
      {"LQRTopology.RejectsDisconnectedTree", [] {
         auto p = five_node_tree();
         p.parents[3] = 4, p.children[3] = 3;
         CHECK(factor_status(p) == Status::INVALID_TOPOLOGY);
       }},
      {"LQRTopology.RejectsCycle", [] {
         auto p = five_node_tree();
         p.parents[0] = 4;
         CHECK(factor_status(p) == Status::INVALID_TOPOLOGY);
       }},
      {"LQRSolve.MatchesDenseKKTOnVariableDimensionTreeProblem", [] {
         auto p = five_node_tree();
         auto input = p.input();
         LQR::Workspace ws;
         ws.reserve(input.dimensions, input.topology);
         {
           auto lqr = LQR(input, ws);
           CHECK(lqr.factor_with_status() == Status::SUCCESS);
           Solution s(p);
           auto out = s.output();
           lqr.solve(out);
           const Vec z = dense_kkt(p);
           // Eigen isApprox(a, b, 1e-10): |a - b| <= 1e-10 min(|a|, |b|)
           auto approx = [](const Vec &a, const double *b) {
             double d = 0, na = 0, nb = 0;
             for (size_t i = 0; i < a.size(); ++i)
               d += (a[i] - b[i]) * (a[i] - b[i]), na += a[i] * a[i], nb += b[i] * b[i];
             return std::sqrt(d) <= 1e-10 * std::sqrt(std::min(na, nb));
           };
           int off = 0;
           for (int node = 0; node < p.N(); ++node) { CHECK(approx(s.x[node], z.data() + off)); off += p.state_dims[node]; }
           for (int e = 0; e < p.E(); ++e) { CHECK(approx(s.u[e], z.data() + off)); off += p.control_dims[e]; }
           for (int node = 0; node < p.N(); ++node) { CHECK(approx(s.y[node], z.data() + off)); off += p.state_dims[node]; }
         }
         ws.free(4);
       }},
      {"Workspace.StaticAndDynamicNumBytesAgree (variable_dimensions_test.cpp:226-263)", [] {
         auto p = default_chain(3, 2, 5);
         auto input = p.input();
         CHECK(LQR::Workspace::num_bytes(3, 2, 5) == LQR::Workspace::num_bytes(input.dimensions, input.topology));
         std::vector<unsigned char> arena(LQR::Workspace::num_bytes(input.dimensions, input.topology));
         LQR::Workspace ws;
         CHECK(ws.mem_assign(input.dimensions, input.topology, arena.data()) == (int)arena.size());
         {
           auto lqr = LQR(input, ws); // arena-backed workspace works end to end
           CHECK(lqr.factor());
           Solution s(p);
           auto out = s.output();
           lqr.solve(out);
           CHECK(kkt_residual(p, s) < 1e-12);
         }
         CHECK(Topology::num_bytes(5) == 40 && Dimensions::num_bytes(5) == (3 * 6 + 15) * 4);
         CHECK(LQR::Output::num_bytes(5) == 17 * (int)sizeof(double *));
       }},
      {"LQR.ZeroDimensionalRootState (variable_dimensions_test.cpp:316-336)", [] {
         Problem p;
         p.parents = {0}, p.children = {1}, p.state_dims = {0, 2}, p.control_dims = {1};
         p.Q = {Mat(0, 0), Mat::identity(2, 1.5)};
         p.q = {{}, {0.3, -0.2}}, p.c = {{}, {0.1, 0.4}}, p.delta = {{}, {0.5, 0.7}};
         p.M = {Mat(0, 1)}, p.R = {Mat::from_rows(1, 1, {1.2})}, p.A = {Mat(2, 0)};
         p.B = {Mat::from_rows(2, 1, {0.6, -0.3})}, p.r = {{0.25}};
         CHECK(solve_residual(p, false) < 1e-12);
       }},
      {"LQR.InputIsDereferencedAtCallTime (helpers.cpp:13-24, 362-368)", [] {
         auto p = nonuniform_diagonal_delta();
         auto full = p.input();
         // construct with null data tables, patch them right before factor/solve
         LQR::Input late{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                         p.dims, p.topo};
         LQR::Workspace ws;
         ws.reserve(3, 2, 3);
         {
           auto lqr = LQR(late, ws);
           late.Q = full.Q, late.M = full.M, late.R = full.R, late.A = full.A, late.B = full.B,
           late.delta = full.delta;
           CHECK(lqr.factor());
           late.q = full.q, late.r = full.r, late.c = full.c;
           Solution s(p);
           auto out = s.output();
           lqr.solve(out);
           CHECK(kkt_residual(p, s) < 1e-12);
         }
         ws.free(3);
       }},
      {"LQR.FusedTreeKernelIsTheDefault (every LQR::Workspace field, against the general engine)", [] {
         // Trees / chains with state dims <= 15, control dims <= 8 run on the fused size-class kernel by default and
         // leave W, K, G_factor, k, V, F_factor, sqrt_delta(_inv), v (lqr.hpp:109-135; read by helpers.cpp:521-665) in
         // the caller's workspace; the same object on the general engine (set_general_engine) must agree on all of
         // them (G_factor / F_factor: the lower triangle, which is all triangularView<Lower> reads).
         for (int which = 0; which < 4; ++which) {
           auto p = which == 0 ? five_node_tree() : which == 1 ? branch_tree() : which == 2 ? variable_dimension_branch()
                                                                                            : nonuniform_diagonal_delta();
           auto input = p.input();
           LQR::Workspace wf, wg;
           wf.reserve(input.dimensions, input.topology);
           wg.reserve(input.dimensions, input.topology);
           {
             auto fused = LQR(input, wf);
             auto general = LQR(input, wg);
             fused.set_general_engine(false); // (whatever SIP_LQR_DROPIN_GENERAL says)
             general.set_general_engine(true);
             CHECK(fused.uses_fused_tree_kernel());
             CHECK(!general.uses_fused_tree_kernel());
             CHECK(fused.factor_with_status() == Status::SUCCESS);
             CHECK(general.factor_with_status() == Status::SUCCESS);
             Solution sf(p), sg(p);
             auto of = sf.output(), og = sg.output();
             fused.solve(of);
             fused.solve(of); // repeatable (lqr_test.cpp:431-450)
             general.solve(og);
             CHECK(kkt_residual(p, sf) < 1e-12);
             double worst = 0.0;
             auto cmp = [&](const double *a, const double *b, int rows, int cols, bool lower_only) {
               double scale = 1.0;
               for (int k = 0; k < rows * cols; ++k)
                 scale = std::fmax(scale, std::fabs(b[k]));
               for (int col = 0; col < cols; ++col)
                 for (int row = lower_only ? col : 0; row < rows; ++row)
                   worst = std::fmax(worst, std::fabs(a[row + rows * col] - b[row + rows * col]) / scale);
             };
             for (int e = 0; e < p.E(); ++e) {
               const int np = p.state_dims[p.parents[e]], nc = p.state_dims[p.children[e]], m = p.control_dims[e];
               cmp(wf.W[e], wg.W[e], nc, nc, false);
               cmp(wf.K[e], wg.K[e], m, np, false);
               cmp(wf.G_factor[e], wg.G_factor[e], m, m, true);
               cmp(wf.k[e], wg.k[e], m, 1, false);
             }
             for (int j = 0; j < p.N(); ++j) {
               const int n = p.state_dims[j];
               cmp(wf.V[j], wg.V[j], n, n, false);
               cmp(wf.F_factor[j], wg.F_factor[j], n, n, true);
               cmp(wf.sqrt_delta[j], wg.sqrt_delta[j], n, 1, false);
               cmp(wf.sqrt_delta_inv[j], wg.sqrt_delta_inv[j], n, 1, false);
               cmp(wf.v[j], wg.v[j], n, 1, false);
             }
             CHECK(worst < 1e-12);
             if (worst >= 1e-12)
               std::printf("    problem %d: worst relative workspace difference %.3e\n", which, worst);
             // statuses through the fused default
             const double keep = p.delta[1][0];
             p.delta[1][0] = 0.0;
             CHECK(fused.factor_with_status() == Status::INVALID_DELTA);
             p.delta[1][0] = keep;
             CHECK(fused.factor_with_status() == Status::SUCCESS);
           }
           wf.free(p.E());
           wg.free(p.E());
         }
       }},
      {"LQR.FusedChainSwitch (adapter addition: uniform chains on the fused kernels, explicit device)", [] {
         // the reference's chain fixture (lqr_test.cpp:229-247) through sip_lqr_factor / sip_lqr_solve:
         // same statuses, same solution (KKT residual < 1e-12), K and k in the caller's workspace
         // satisfy u = K x + k (lqr.cpp:856-857); trees ignore the switch
         auto p = nonuniform_diagonal_delta();
         auto input = p.input();
         LQR::Workspace ws;
         ws.reserve(3, 2, 3);
         {
           auto lqr = LQR(input, ws);
           lqr.set_device(0);
           lqr.set_fused_chains(true);
           CHECK(lqr.uses_fused_chain_kernel());
           CHECK(lqr.factor_with_status() == Status::SUCCESS);
           Solution s(p);
           auto out = s.output();
           lqr.solve(out);
           lqr.solve(out); // repeatable (lqr_test.cpp:431-450)
           CHECK(kkt_residual(p, s) < 1e-12);
           double worst = 0.0;
           for (int e = 0; e < 3; ++e)
             for (int j = 0; j < 2; ++j) {
               double pred = ws.k[e][j];
               for (int col = 0; col < 3; ++col)
                 pred += ws.K[e][j + 2 * col] * s.x[e][col];
               worst = std::fmax(worst, std::fabs(pred - s.u[e][j]));
             }
           CHECK(worst < 1e-12);
           p.delta[2][0] = 0.0;
           CHECK(lqr.factor_with_status() == Status::INVALID_DELTA);
           p.delta[2][0] = 0.3;
         }
         ws.free(3);
         auto tree = branch_tree();
         auto tin = tree.input();
         LQR::Workspace tws;
         tws.reserve(2, 1, 2);
         {
           auto lqr = LQR(tin, tws);
           lqr.set_fused_chains(true);
           CHECK(!lqr.uses_fused_chain_kernel());
           CHECK(lqr.factor());
         }
         tws.free(2);
       }},
  };
  for (auto &c : cases) {
    const int before = g_failures;
    c.run();
    std::printf("[%s] %s\n", g_failures == before ? "  OK  " : "FAILED", c.name);
  }
  std::printf("%d checks, %d failures\n", g_checks, g_failures);
  return g_failures == 0 ? 0 : 1;
}
