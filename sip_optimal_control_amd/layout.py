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

    # scalars per stage block
    @property
    def node(self):  # Q | delta
        return self.n * self.n + self.n

    @property
    def edge(self):  # A | B | M | R
        return self.n * self.n + 2 * self.n * self.m + self.m * self.m

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
        off = {"Q": base, "delta": base + n * n}
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
