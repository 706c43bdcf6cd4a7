"""The hard-coded problems of the reference's own hot-path tests, as data.

Re-expressed (not copied: these are the numeric fixtures, in numpy form) from
/root/reference/tests/lqr_test.cpp.  Matrices are numpy (row, col) arrays; the
Eigen comma initialisers of the reference fill row-major, which is what the
nested lists below spell out.

Each builder returns dict(parents, children, state_dims, control_dims, blocks)
with blocks[name] = list of arrays, name in Q M R q r A B c delta.
"""
import numpy as np


def _problem(parents, children, state_dims, control_dims, blocks):
    return dict(parents=list(parents), children=list(children), state_dims=list(state_dims),
                control_dims=list(control_dims), blocks=blocks)


def default_chain(n, m, T):
    """LQRProblem(n, m, T) constructor defaults, lqr_test.cpp:45-75:
    Q = I, q = 0, c = 0, delta = 1, M = 0, R = I, A = I, B = ones, r = 0."""
    blocks = {
        "Q": [np.eye(n) for _ in range(T + 1)],
        "q": [np.zeros(n) for _ in range(T + 1)],
        "c": [np.zeros(n) for _ in range(T + 1)],
        "delta": [np.ones(n) for _ in range(T + 1)],
        "M": [np.zeros((n, m)) for _ in range(T)],
        "R": [np.eye(m) for _ in range(T)],
        "A": [np.eye(n) for _ in range(T)],
        "B": [np.ones((n, m)) for _ in range(T)],
        "r": [np.zeros(m) for _ in range(T)],
    }
    return _problem(range(T), range(1, T + 1), [n] * (T + 1), [m] * T, blocks)


def nonuniform_diagonal_delta():
    """LQRSolve.SolvesNonuniformDiagonalDeltaProblem, lqr_test.cpp:229-247."""
    prob = default_chain(3, 2, 3)
    b = prob["blocks"]
    T = 3
    for i in range(T):
        b["A"][i] = np.array([[1.0 + 0.02 * i, 0.03, -0.01],
                              [-0.02, 0.95 + 0.01 * i, 0.04],
                              [0.01, -0.03, 1.02 - 0.01 * i]])
        b["B"][i] = np.array([[0.2, -0.1], [0.05, 0.15], [-0.1, 0.08]])
        b["Q"][i] = np.diag([1.0 + 0.1 * i, 1.4 + 0.05 * i, 1.8 + 0.03 * i])
        b["R"][i] = np.diag([1.2 + 0.1 * i, 1.6 + 0.07 * i])
        b["q"][i] = np.array([0.2 + 0.01 * i, -0.1 + 0.02 * i, 0.05 - 0.03 * i])
        b["r"][i] = np.array([-0.2 + 0.03 * i, 0.1 - 0.01 * i])
        b["c"][i] = np.array([0.03 + 0.01 * i, -0.04 + 0.02 * i, 0.02 - 0.01 * i])
        b["delta"][i] = np.array([0.03 + 0.01 * i, 0.11 + 0.02 * i, 0.19 + 0.03 * i])
    b["Q"][T] = np.diag([1.3, 1.7, 2.1])
    b["q"][T] = np.array([0.06, -0.08, 0.12])
    b["c"][T] = np.array([-0.02, 0.05, -0.01])
    b["delta"][T] = np.array([0.07, 0.17, 0.29])
    return prob


def branch_tree():
    """BranchLQRProblem, lqr_test.cpp:300-335: 3-node star, parents {0,0}."""
    blocks = {
        "Q": [np.array([[2.0, 0.1], [0.1, 1.5]]), np.array([[1.3, 0.2], [0.2, 1.7]]),
              np.array([[1.8, -0.1], [-0.1, 1.4]])],
        "M": [np.array([[0.2], [-0.1]]), np.array([[-0.15], [0.05]])],
        "R": [np.array([[1.6]]), np.array([[1.9]])],
        "A": [np.array([[1.0, 0.2], [0.0, 0.9]]), np.array([[0.8, -0.1], [0.3, 1.1]])],
        "B": [np.array([[0.4], [0.2]]), np.array([[-0.1], [0.5]])],
        "q": [np.array([0.3, -0.2]), np.array([-0.1, 0.4]), np.array([0.2, 0.1])],
        "r": [np.array([-0.3]), np.array([0.25])],
        "c": [np.array([0.1, -0.2]), np.array([-0.05, 0.1]), np.array([0.2, 0.15])],
        "delta": [np.array([0.7, 0.9]), np.array([0.8, 1.1]), np.array([1.0, 0.6])],
    }
    return _problem([0, 0], [1, 2], [2, 2, 2], [1, 1], blocks)


def variable_dimension_branch():
    """VariableDimensionBranchProblem, lqr_test.cpp:494-532."""
    blocks = {
        "Q": [np.array([[2.0, 0.1], [0.1, 1.7]]), np.array([[1.3]]),
              np.array([[1.8, 0.1, -0.2], [0.1, 1.6, 0.05], [-0.2, 0.05, 2.1]])],
        "M": [np.array([[0.1, -0.2], [0.05, 0.15]]), np.array([[-0.1], [0.2]])],
        "R": [np.array([[1.8, 0.1], [0.1, 1.5]]), np.array([[1.4]])],
        "A": [np.array([[0.8, -0.3]]), np.array([[1.0, 0.2], [-0.1, 0.7], [0.3, -0.4]])],
        "B": [np.array([[0.4, -0.2]]), np.array([[0.2], [-0.1], [0.5]])],
        "q": [np.array([0.2, -0.15]), np.array([-0.05]), np.array([0.1, -0.2, 0.05])],
        "r": [np.array([-0.1, 0.25]), np.array([-0.2])],
        "c": [np.array([0.05, -0.1]), np.array([0.12]), np.array([-0.02, 0.04, -0.08])],
        "delta": [np.array([0.8, 1.1]), np.array([0.9]), np.array([0.7, 1.0, 1.2])],
    }
    return _problem([0, 0], [1, 2], [2, 1, 3], [2, 1], blocks)


def five_node_variable_tree():
    """FiveNodeVariableTreeProblem, lqr_test.cpp:661-762 (closed-form data)."""
    parents, children = [0, 0, 1, 1], [1, 2, 3, 4]
    state_dims, control_dims = [3, 1, 2, 4, 2], [2, 1, 3, 1]
    blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
    for node, n in enumerate(state_dims):
        Q = np.eye(n) * (1.5 + 0.2 * node)
        for col in range(n):
            for row in range(col + 1, n):
                Q[row, col] = Q[col, row] = 0.02 * (row + col + node + 1)
        blocks["Q"].append(Q)
        blocks["q"].append(np.linspace(-0.15 + 0.03 * node, 0.12 + 0.02 * node, n))
        blocks["c"].append(np.linspace(0.05 * node, 0.04 + 0.03 * node, n))
        blocks["delta"].append(np.linspace(0.7 + 0.05 * node, 1.0 + 0.04 * node, n))
    for e, m in enumerate(control_dims):
        n_parent, n_child = state_dims[parents[e]], state_dims[children[e]]
        M = np.array([[0.015 * ((e + 1) * (row + 1) - col) for col in range(m)]
                      for row in range(n_parent)], dtype=float).reshape(n_parent, m)
        A = np.array([[0.08 * (row + 1) / (e + col + 2) for col in range(n_parent)]
                      for row in range(n_child)], dtype=float).reshape(n_child, n_parent)
        B = np.array([[-0.06 * (col + 1) / (e + row + 2) for col in range(m)]
                      for row in range(n_child)], dtype=float).reshape(n_child, m)
        R = np.eye(m) * (1.8 + 0.1 * e)
        for col in range(m):
            for row in range(col + 1, m):
                R[row, col] = R[col, row] = 0.03 * (row + col + 1)
        blocks["M"].append(M)
        blocks["A"].append(A)
        blocks["B"].append(B)
        blocks["R"].append(R)
        blocks["r"].append(np.linspace(-0.2 + 0.04 * e, 0.1 + 0.03 * e, m))
    return _problem(parents, children, state_dims, control_dims, blocks)


# Eigen::VectorXd::LinSpaced(1, lo, hi) returns [hi]; numpy.linspace(lo, hi, 1)
# returns [lo].  The five-node problem has size-1 blocks (node 1, edges 1 and 3).
def _eigen_linspaced(n, lo, hi):
    return np.array([hi]) if n == 1 else np.linspace(lo, hi, n)


def five_node_variable_tree_eigen():
    prob = five_node_variable_tree()
    b = prob["blocks"]
    for node, n in enumerate(prob["state_dims"]):
        b["q"][node] = _eigen_linspaced(n, -0.15 + 0.03 * node, 0.12 + 0.02 * node)
        b["c"][node] = _eigen_linspaced(n, 0.05 * node, 0.04 + 0.03 * node)
        b["delta"][node] = _eigen_linspaced(n, 0.7 + 0.05 * node, 1.0 + 0.04 * node)
    for e, m in enumerate(prob["control_dims"]):
        b["r"][e] = _eigen_linspaced(m, -0.2 + 0.04 * e, 0.1 + 0.03 * e)
    return prob


def variable_benchmark_problem(shape, T, base_n, base_m, rng):
    """VariableLQRProblem of the reference's benchmark (benchmarks/lqr_benchmark.cpp:209-310):
    shape 0 heterogeneous chain, 1 shallow wide tree (every edge leaves the root), 2 binary tree;
    state_dims[node] = max(1, base_n + node % 3 - 1), control_dims[edge] = max(1, base_m + edge % 3 - 1);
    A = 0.05 N(0,1), B = 0.1 N(0,1), M = 0, R = G^T G + I, Q = S^T S + 1e-3 I, q, r, c ~ N(0,1),
    delta = 1e-3 + 0.1 U(0,1) (the value generator is numpy's: std::normal_distribution is
    implementation-defined)."""
    sd = [max(1, base_n + (node % 3) - 1) for node in range(T + 1)]
    cd = [max(1, base_m + (edge % 3) - 1) for edge in range(T)]
    children = list(range(1, T + 1))
    parents = [{0: e, 1: 0, 2: (e + 1 - 1) // 2}[shape] for e in range(T)]
    blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
    for n in sd:
        S = rng.normal(size=(n, n))
        blocks["Q"].append(S.T @ S + 1e-3 * np.eye(n))
        blocks["q"].append(rng.normal(size=n))
        blocks["c"].append(rng.normal(size=n))
        blocks["delta"].append(1e-3 + 0.1 * rng.random(n))
    for e, m in enumerate(cd):
        np_, nc = sd[parents[e]], sd[children[e]]
        G = rng.normal(size=(m, m))
        blocks["A"].append(0.05 * rng.normal(size=(nc, np_)))
        blocks["B"].append(0.1 * rng.normal(size=(nc, m)))
        blocks["M"].append(np.zeros((np_, m)))
        blocks["R"].append(G.T @ G + np.eye(m))
        blocks["r"].append(rng.normal(size=m))
    return _problem(parents, children, sd, cd, blocks)
