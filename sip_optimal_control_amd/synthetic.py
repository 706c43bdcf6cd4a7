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
