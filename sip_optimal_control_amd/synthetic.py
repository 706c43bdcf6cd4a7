"""Synthetic stagewise LQR data: the recipe of the reference's benchmark
(benchmarks/lqr_benchmark.cpp:61-98), batched.

  A = I + 0.05 N(0,1), B = 0.1 N(0,1), M = cross_term * N(0,1) (reference: 0),
  R = G^T G + 1.01 I, Q = S^T S + 1e-3 I (terminal too), q, r, c ~ N(0,1),
  delta = 1e-3 + 0.1 U(0,1).

The reference seeds std::mt19937(0) and draws through std::normal_distribution,
whose output is standard-library specific, so bit-identical inputs cannot be
reproduced anyway; here every batch element is an independent draw from one
seeded torch generator (same distributions).  Returns (mats, vecs) in the
packed chain layout on `device`.
"""
import torch

from .layout import ChainShape


def make_chain_batch(shape: ChainShape, batch, seed=0, device="cpu", dtype=torch.float64,
                     cross_term=0.0, chunk=512):
    n, m, T = shape.n, shape.m, shape.T
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    mats = torch.empty(batch, shape.mats_len, dtype=dtype, device=dev)
    vecs = torch.empty(batch, shape.vecs_len, dtype=dtype, device=dev)
    stg, vstg = shape.node + shape.edge, shape.vnode + shape.vedge
    f64 = torch.float64

    def randn(*size):
        return torch.randn(*size, generator=gen, device=dev, dtype=f64)

    def rand(*size):
        return torch.rand(*size, generator=gen, device=dev, dtype=f64)

    eye_n = torch.eye(n, device=dev, dtype=f64)
    eye_m = torch.eye(m, device=dev, dtype=f64)
    for lo in range(0, batch, chunk):
        hi = min(batch, lo + chunk)
        b = hi - lo
        # nodes 0..T
        S = randn(b, T + 1, n, n)
        Q = S.transpose(-1, -2) @ S + 1e-3 * eye_n
        delta = 1e-3 + 0.1 * rand(b, T + 1, n)
        q = randn(b, T + 1, n)
        c = randn(b, T + 1, n)
        node = torch.cat([Q.reshape(b, T + 1, n * n), delta], dim=-1)
        vnode = torch.cat([q, c], dim=-1)
        mats[lo:hi, T * stg:] = node[:, T].to(dtype)
        vecs[lo:hi, T * vstg:] = vnode[:, T].to(dtype)
        if T > 0:
            A = eye_n + 0.05 * randn(b, T, n, n)
            B = 0.1 * randn(b, T, m, n)      # storage order: column j of B contiguous
            Mx = cross_term * randn(b, T, m, n)
            G = randn(b, T, m, m)
            R = G.transpose(-1, -2) @ G + 1.01 * eye_m
            r = randn(b, T, m)
            edge = torch.cat([A.reshape(b, T, n * n), B.reshape(b, T, n * m),
                              Mx.reshape(b, T, n * m), R.reshape(b, T, m * m)], dim=-1)
            mats[lo:hi, :T * stg] = torch.cat([node[:, :T], edge], dim=-1).reshape(b, T * stg).to(dtype)
            vecs[lo:hi, :T * vstg] = torch.cat([vnode[:, :T], r], dim=-1).reshape(b, T * vstg).to(dtype)
    return mats, vecs


def make_newton_kkt_batch(kkt, state_dims, control_dims, parents, children, node_c_dims, node_g_dims,
                          edge_c_dims, edge_g_dims, seed=0, r2_max=1e9):
    """Model-callback outputs, regularization and right-hand sides with the value
    distributions of the reference's Newton-KKT benchmark
    (benchmarks/newton_kkt_benchmark.cpp:170-262), batched on the device of
    `kkt` (a BatchedNewtonKKT):  dc/dg Jacobians 0.1 N(0,1), node d2L_dx2 =
    S^T S + 1e-3 I, ddyn_dx = I + 0.05 N(0,1), ddyn_du = 0.1 N(0,1), edge
    d2L_dx2 = 0, d2L_dxdu = 0.01 N(0,1), d2L_du2 = G^T G + I, r2 log-uniform
    [1e-3, r2_max], w log-uniform [1e-2, 1e3], r3 log-uniform [1e-3, 1e1],
    r1 = 1e-8, rhs ~ N(0,1).  Returns (model, w, r1, r2, r3, rhs)."""
    import math
    dev = kkt.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    f64 = torch.float64
    B = kkt.batch

    def randn(*size):
        return torch.randn(B, *size, generator=gen, device=dev, dtype=f64)

    def logu(lo, hi, length):
        u = torch.rand(B, length, generator=gen, device=dev, dtype=f64)
        return torch.exp(math.log(lo) + (math.log(hi) - math.log(lo)) * u)

    def spd(k, shift):
        root = randn(k, k)
        return root.transpose(-1, -2) @ root + shift * torch.eye(k, device=dev, dtype=f64)

    model = torch.zeros(B, max(1, kkt.model_len), dtype=f64, device=dev)

    def put(block, index, mat):  # mat: [B, rows, cols] -> column-major block
        if mat.shape[1] * mat.shape[2] == 0:
            return
        off = kkt.model_offset(block, index)
        model[:, off:off + mat.shape[1] * mat.shape[2]] = mat.transpose(-1, -2).reshape(B, -1)

    N, E = len(state_dims), len(control_dims)
    for i in range(N):
        n, c, g = state_dims[i], node_c_dims[i], node_g_dims[i]
        put(0, i, spd(n, 1e-3))
        put(1, i, 0.1 * randn(c, n))
        put(2, i, 0.1 * randn(g, n))
    for e in range(E):
        np_, nc, m = state_dims[parents[e]], state_dims[children[e]], control_dims[e]
        c, g = edge_c_dims[e], edge_g_dims[e]
        put(4, e, 0.01 * randn(np_, m))
        put(5, e, spd(m, 1.0))
        put(6, e, torch.eye(nc, np_, device=dev, dtype=f64) + 0.05 * randn(nc, np_))
        put(7, e, 0.1 * randn(nc, m))
        put(8, e, 0.1 * randn(c, np_))
        put(9, e, 0.1 * randn(c, m))
        put(10, e, 0.1 * randn(g, np_))
        put(11, e, 0.1 * randn(g, m))
    r2 = logu(1e-3, r2_max, kkt.y_dim)
    w = logu(1e-2, 1e3, kkt.z_dim)
    r3 = logu(1e-3, 1e1, kkt.z_dim)
    r1 = torch.full((B, kkt.x_dim), 1e-8, dtype=f64, device=dev)
    rhs = randn(kkt.kkt_dim)
    return model, w, r1, r2, r3, rhs
