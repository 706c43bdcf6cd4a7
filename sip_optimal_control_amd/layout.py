"""Packed chain layout (include/sip_lqr_amd.h, "Packed chain layout").

Host-side index arithmetic shared by the pack helpers, the synthetic problem
generator and the tests.  n = state dim, m = control dim, T = num_edges.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class ChainShape:
    n: int
    m: int
    T: int
    # SIP_LQR_LAYOUT_SYMMETRIC: Q and R as lower triangles packed by columns (include/sip_lqr_amd.h)
    symmetric: bool = False

    @property
    def qlen(self):
        return self.n * (self.n + 1) // 2 if self.symmetric else self.n * self.n

    @property
    def rlen(self):
        return self.m * (self.m + 1) // 2 if self.symmetric else self.m * self.m

    # scalars per stage block
    @property
    def node(self):  # Q | delta
        return self.qlen + self.n

    @property
    def edge(self):  # A | B | M | R
        return self.n * self.n + 2 * self.n * self.m + self.rlen

    def full(self):
        return ChainShape(self.n, self.m, self.T)

    def packed(self):
        return ChainShape(self.n, self.m, self.T, symmetric=True)

    def pack_index(self):
        """Index array (numpy int64): mats of the FULL layout [.., full.mats_len] -> mats of this (symmetric)
        layout, `sym = full[..., index]`."""
        import numpy as np
        f = self.full()
        assert self.symmetric
        n, m = self.n, self.m
        tri_n = [r + n * c for c in range(n) for r in range(c, n)]
        tri_m = [r + m * c for c in range(m) for r in range(c, m)]
        idx = []
        for i in range(self.T + 1):
            o = f.mats_off(i)
            idx += [o["Q"] + k for k in tri_n] + [o["delta"] + k for k in range(n)]
            if i < self.T:
                idx += list(range(o["A"], o["R"])) + [o["R"] + k for k in tri_m]
        return np.asarray(idx, dtype=np.int64)

    @property
    def vnode(self):  # q | c   (x | y in sol)
        return 2 * self.n

    @property
    def vedge(self):  # r (u in sol)
        return self.m

    @property
    def gain(self):  # K | k
        return self.m * self.n + self.m

    @property
    def mats_len(self):
        return (self.T + 1) * self.node + self.T * self.edge

    @property
    def vecs_len(self):
        return (self.T + 1) * self.vnode + self.T * self.vedge

    @property
    def gains_len(self):
        return self.T * self.gain

    # offsets of the blocks of stage i inside one problem
    def mats_off(self, i):
        base = i * (self.node + self.edge)
        n, m = self.n, self.m
        off = {"Q": base, "delta": base + self.qlen}
        e = base + self.node
        off.update({"A": e, "B": e + n * n, "M": e + n * n + n * m, "R": e + n * n + 2 * n * m})
        return off

    def vecs_off(self, i):
        base = i * (self.vnode + self.vedge)
        return {"q": base, "c": base + self.n, "r": base + 2 * self.n}

    def sol_off(self, i):
        base = i * (self.vnode + self.vedge)
        return {"x": base, "y": base + self.n, "u": base + 2 * self.n}

    def gains_off(self, i):
        base = i * self.gain
        return {"K": base, "k": base + self.m * self.n}

    # compulsory bytes and flops of one sweep
    def algorithmic_bytes(self, scalar_bytes=8):
        n, m, T = self.n, self.m, self.T
        inputs = (T + 1) * (n * n + 3 * n) + T * (n * n + 2 * n * m + m * m + m)
        outputs = 2 * (T + 1) * n + T * m + T * (m * n + m)
        return scalar_bytes * (inputs + outputs)

    def algorithmic_flops(self):
        n, m, T = self.n, self.m, self.T
        return T * (6.33 * n ** 3 + 6 * m * n * n + 4 * m * m * n + m ** 3 / 3.0
                    + 10 * n * n + 8 * m * n + 2 * m * m)
