"""Model data of the reference's own Newton-KKT tests, as data.

Re-expressed from /root/reference/tests/variable_dimensions_test.cpp
(`initialize_model` :71-133, `expect_kkt_solve` :135-181 and the dimension
tables of the three theta_dim == 0 CallbackProvider tests :265-336) and the
value distributions of benchmarks/newton_kkt_benchmark.cpp:58-262 (the
generator is ours: std::normal_distribution is implementation-defined).
"""
import numpy as np

from oracle.kkt import KKTDims


def _sequence(shape, scale):
    """fill_sequence (:46-50): storage element i = scale * (i + 1), column-major."""
    rows, cols = shape
    return (scale * np.arange(1, rows * cols + 1, dtype=np.float64)).reshape((rows, cols), order="F")


def initialize_model(dims):
    nodes, edges = [], []
    for i in range(dims.N):
        n, c, g = dims.sd[i], dims.ncd[i], dims.ngd[i]
        nodes.append({"d2L_dx2": (2.5 + 0.2 * i) * np.eye(n),
                      "dc_dx": _sequence((c, n), 0.013 * (i + 1)),
                      "dg_dx": _sequence((g, n), -0.011 * (i + 1))})
    for e in range(dims.E):
        np_, nc, m = dims.sd[dims.parents[e]], dims.sd[dims.children[e]], dims.cd[e]
        c, g = dims.ecd[e], dims.egd[e]
        edges.append({"d2L_dx2": (0.3 + 0.05 * e) * np.eye(np_),
                      "d2L_dxdu": _sequence((np_, m), 0.009 * (e + 1)),
                      "d2L_du2": (3.0 + 0.2 * e) * np.eye(m),
                      "ddyn_dx": _sequence((nc, np_), 0.025 + 0.004 * e),
                      "ddyn_du": _sequence((nc, m), -0.031 - 0.003 * e),
                      "dc_dx": _sequence((c, np_), 0.017 * (e + 1)),
                      "dc_du": _sequence((c, m), 0.019 * (e + 1)),
                      "dg_dx": _sequence((g, np_), -0.014 * (e + 1)),
                      "dg_du": _sequence((g, m), 0.016 * (e + 1))})
    return dims.pack_model(nodes, edges)


def initialize_theta_model(dims, theta_diagonal):
    """The theta blocks of initialize_model (:93-98, 119-131)."""
    p = dims.p
    nodes, edges = [], []
    for i in range(dims.N):
        n, c, g = dims.sd[i], dims.ncd[i], dims.ngd[i]
        nodes.append({"d2L_dxdtheta": _sequence((n, p), 0.0005 * (i + 1)),
                      "dc_dtheta": _sequence((c, p), 0.001 * (i + 1)),
                      "dg_dtheta": _sequence((g, p), -0.0007 * (i + 1)),
                      "d2L_dtheta2": theta_diagonal * np.eye(p)})
    for e in range(dims.E):
        np_, nc, m = dims.sd[dims.parents[e]], dims.sd[dims.children[e]], dims.cd[e]
        edges.append({"d2L_dxdtheta": _sequence((np_, p), 0.0004 * (e + 1)),
                      "d2L_dudtheta": _sequence((m, p), -0.0003 * (e + 1)),
                      "ddyn_dtheta": _sequence((nc, p), 0.0009 * (e + 1)),
                      "dc_dtheta": _sequence((dims.ecd[e], p), 0.0008 * (e + 1)),
                      "dg_dtheta": _sequence((dims.egd[e], p), -0.0006 * (e + 1)),
                      "d2L_dtheta2": theta_diagonal * np.eye(p)})
    return dims.pack_theta(nodes, edges)


def regularization(dims):
    """expect_kkt_solve, :144-151 and the rhs of :155-156 (x_dim includes theta)."""
    w = np.full(dims.z_dim, 1.3)
    r2 = np.full(dims.y_dim, 0.9)
    r3 = np.full(dims.z_dim, 0.4)
    r1 = 0.03 * np.arange(1, dims.x_dim + dims.p + 1) + 0.2
    rhs = 0.01 * np.arange(1, dims.full_dim + 1)
    return w, r1, r2, r3, rhs


CHAIN = dict(parents=[0, 1], children=[1, 2])
BRANCH = dict(parents=[0, 0], children=[1, 2])

REFERENCE_CASES = {
    # CallbackProvider.SolvesChainWithNodeAndEdgeConstraints, :265-288
    "chain_node_edge_constraints": dict(CHAIN, state_dims=[2, 1, 3], control_dims=[1, 2], node_c=[1, 0, 2],
                                        node_g=[0, 2, 1], edge_c=[1, 2], edge_g=[2, 1]),
    # CallbackProvider.SolvesIndependentConstraintsOnSiblingEdges, :290-313
    "branch_sibling_edges": dict(BRANCH, state_dims=[2, 1, 3], control_dims=[1, 2], node_c=[1, 0, 1],
                                 node_g=[1, 1, 0], edge_c=[2, 1], edge_g=[1, 2]),
    # CallbackProvider.SolvesBranchedSystemWithZeroDimensionalRoot, :315-336
    "branch_zero_dim_root": dict(BRANCH, state_dims=[0, 1, 3], control_dims=[1, 2], node_c=[0, 0, 0],
                                 node_g=[0, 0, 0], edge_c=[0, 0], edge_g=[0, 0]),
}


# CallbackProvider.SolvesBranchedSystemWithSchurVariables, :338-363 (theta_dim = 2, tolerance 1e-8)
SCHUR_CASE = dict(BRANCH, state_dims=[2, 1, 3], control_dims=[1, 2], node_c=[1, 0, 1], node_g=[0, 1, 1],
                  edge_c=[1, 2], edge_g=[2, 1], theta_dim=2)


def schur_case():
    dims = KKTDims(**SCHUR_CASE)
    return dims, initialize_model(dims), initialize_theta_model(dims, 6.0), regularization(dims)


def reference_case(name):
    dims = KKTDims(**REFERENCE_CASES[name])
    return dims, initialize_model(dims), regularization(dims)


def newton_kkt_dims(n, m, T):
    """NewtonKKTProblem(n, m, T), newton_kkt_benchmark.cpp:58-83: c = max(1, n/2)
    equality and g = max(1, 2m) inequality rows per edge, and on the last node."""
    c, g = max(1, n // 2), max(1, 2 * m)
    return KKTDims(parents=list(range(T)), children=list(range(1, T + 1)), state_dims=[n] * (T + 1),
                   control_dims=[m] * T, node_c=[0] * T + [c], node_g=[0] * T + [g], edge_c=[c] * T,
                   edge_g=[g] * T)


def newton_kkt_problem(dims, seed, batch=None, r2_max=1e9):
    """Value distributions of newton_kkt_benchmark.cpp:170-262; returns
    (model, w, r1, r2, r3, rhs), each [batch, len] (or 1-D if batch is None)."""
    rng = np.random.default_rng(seed)
    count = 1 if batch is None else batch

    def spd(k, shift):
        root = rng.standard_normal((k, k))
        return root.T @ root + shift * np.eye(k)

    models = []
    for _ in range(count):
        nodes, edges = [], []
        for i in range(dims.N):
            n, c, g = dims.sd[i], dims.ncd[i], dims.ngd[i]
            nodes.append({"dc_dx": 0.1 * rng.standard_normal((c, n)), "dg_dx": 0.1 * rng.standard_normal((g, n)),
                          "d2L_dx2": spd(n, 1e-3)})
        for e in range(dims.E):
            np_, nc, m = dims.sd[dims.parents[e]], dims.sd[dims.children[e]], dims.cd[e]
            c, g = dims.ecd[e], dims.egd[e]
            edges.append({"ddyn_dx": np.eye(nc, np_) + 0.05 * rng.standard_normal((nc, np_)),
                          "ddyn_du": 0.1 * rng.standard_normal((nc, m)),
                          "dc_dx": 0.1 * rng.standard_normal((c, np_)), "dc_du": 0.1 * rng.standard_normal((c, m)),
                          "dg_dx": 0.1 * rng.standard_normal((g, np_)), "dg_du": 0.1 * rng.standard_normal((g, m)),
                          "d2L_dx2": np.zeros((np_, np_)), "d2L_dxdu": 0.01 * rng.standard_normal((np_, m)),
                          "d2L_du2": spd(m, 1.0)})
        models.append(dims.pack_model(nodes, edges))
    model = np.stack(models)
    thetas = []
    if dims.p > 0:  # theta blocks of :184-236: 1e-3 N(0,1) couplings, d2L_dtheta2 = G^T G + 100 I on the last node
        pp = dims.p
        for _ in range(count):
            tn = [{"d2L_dxdtheta": 1e-3 * rng.standard_normal((dims.sd[i], pp)),
                   "dc_dtheta": 1e-3 * rng.standard_normal((dims.ncd[i], pp)),
                   "dg_dtheta": 1e-3 * rng.standard_normal((dims.ngd[i], pp)),
                   "d2L_dtheta2": spd(pp, 100.0) if i == dims.E else np.zeros((pp, pp))} for i in range(dims.N)]
            te = [{"d2L_dxdtheta": 1e-3 * rng.standard_normal((dims.sd[dims.parents[e]], pp)),
                   "d2L_dudtheta": 1e-3 * rng.standard_normal((dims.cd[e], pp)),
                   "ddyn_dtheta": 1e-3 * rng.standard_normal((dims.sd[dims.children[e]], pp)),
                   "dc_dtheta": 1e-3 * rng.standard_normal((dims.ecd[e], pp)),
                   "dg_dtheta": 1e-3 * rng.standard_normal((dims.egd[e], pp)),
                   "d2L_dtheta2": np.zeros((pp, pp))} for e in range(dims.E)]
            thetas.append(dims.pack_theta(tn, te))
    logu = lambda lo, hi, size: np.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * rng.random(size))
    r2 = logu(1e-3, r2_max, (count, dims.y_dim))
    w = logu(1e-2, 1e3, (count, dims.z_dim))
    r3 = logu(1e-3, 1e1, (count, dims.z_dim))
    r1 = np.full((count, dims.x_dim + dims.p), 1e-8)
    rhs = rng.standard_normal((count, dims.full_dim))
    out = (model, w, r1, r2, r3, rhs) + ((np.stack(thetas),) if dims.p > 0 else ())
    return out if batch is not None else tuple(a[0] for a in out)
