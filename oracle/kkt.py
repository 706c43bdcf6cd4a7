"""ctypes front-end of oracle/kkt_oracle.c and an independent dense KKT check.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

`KKTOracle` mirrors `CallbackProvider` (helpers.hpp:7-33) for theta_dim == 0 on
the flat model arena documented in kkt_oracle.h.  `dense_kkt_matrix` assembles

    K = [[H + diag(r1), C^T, G^T], [C, -diag(r2), 0], [G, 0, -diag(w + r3)]]

directly from the model blocks with numpy (nothing shared with the C code), in
the variable ordering of types.cpp:24-64; `numpy.linalg.solve(K, rhs)` is the
independent answer that the condensation + Riccati + recovery path must match.
"""
import ctypes

import numpy as np

from . import oracle as _oracle

THETA_NODE_BLOCKS = ("d2L_dxdtheta", "dc_dtheta", "dg_dtheta", "d2L_dtheta2")
THETA_EDGE_BLOCKS = ("d2L_dxdtheta", "d2L_dudtheta", "ddyn_dtheta", "dc_dtheta", "dg_dtheta", "d2L_dtheta2")
NODE_BLOCKS = ("d2L_dx2", "dc_dx", "dg_dx")
EDGE_BLOCKS = ("d2L_dx2", "d2L_dxdu", "d2L_du2", "ddyn_dx", "ddyn_du", "dc_dx", "dc_du", "dg_dx", "dg_du")
_TABLES = ("x_state", "x_control", "y_dyn", "y_node_c", "y_edge_c", "z_node", "z_edge")
STATUS_NAMES = dict(_oracle.STATUS_NAMES)
STATUS_NAMES.update({5: "NONPOSITIVE_REGULARIZATION", 6: "INVALID_INPUT", 7: "THETA_SCHUR_FAILURE"})

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)
_ready = False


def _lib():
    global _ready
    L = _oracle.lib()
    if not _ready:
        L.kkt_oracle_create.argtypes = [ctypes.c_int, ctypes.c_int] + [_I] * 8
        L.kkt_oracle_create.restype = ctypes.c_void_p
        L.kkt_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.kkt_oracle_destroy.restype = None
        L.kkt_oracle_dim.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.kkt_oracle_dim.restype = ctypes.c_long
        for name in ("kkt_oracle_model_offset", "kkt_oracle_vector_offset"):
            fn = getattr(L, name)
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            fn.restype = ctypes.c_long
        L.kkt_oracle_factor.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        L.kkt_oracle_solve.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
        L.kkt_oracle_solve.restype = None
        L.kkt_oracle_add_Kx_to_y.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 7
        L.kkt_oracle_add_Kx_to_y.restype = None
        L.kkt_oracle_lqr_block.argtypes = [ctypes.c_void_p, ctypes.c_char, ctypes.c_int]
        L.kkt_oracle_lqr_block.restype = _D
        L.kkt_oracle_set_theta.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.kkt_oracle_set_theta.restype = None
        L.kkt_oracle_theta_len.argtypes = [ctypes.c_void_p]
        L.kkt_oracle_theta_len.restype = ctypes.c_long
        L.kkt_oracle_theta_offset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.kkt_oracle_theta_offset.restype = ctypes.c_long
        L.kkt_oracle_factor_theta.argtypes = [ctypes.c_void_p] * 7
        L.kkt_oracle_solve_theta.argtypes = [ctypes.c_void_p] * 5
        L.kkt_oracle_solve_theta.restype = None
        L.kkt_oracle_add_Kx_to_y_theta.argtypes = [ctypes.c_void_p] * 9
        L.kkt_oracle_add_Kx_to_y_theta.restype = None
        for name in ("Hx", "Cx", "CTx", "Gx", "GTx"):
            fn = getattr(L, f"kkt_oracle_add_{name}_to_y")
            fn.argtypes = [ctypes.c_void_p] * 5
            fn.restype = None
        L.kkt_oracle_batch.argtypes = [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_void_p] * 8 + [ctypes.c_int]
        _ready = True
    return L


def _ints(values):
    if values is None:
        return None
    return (ctypes.c_int * max(1, len(values)))(*[int(v) for v in values])


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data


class KKTDims:
    """Dimension tables + the reference's flattened offsets, computed in Python
    (types.cpp:24-64) so tests can cross-check the C and HIP tables."""

    def __init__(self, parents, children, state_dims, control_dims, node_c=None, node_g=None,
                 edge_c=None, edge_g=None, root=0, theta_dim=0):
        self.p = int(theta_dim)
        self.parents, self.children = list(parents), list(children)
        self.sd, self.cd = list(state_dims), list(control_dims)
        self.E, self.N, self.root = len(self.cd), len(self.cd) + 1, root
        z = lambda v, k: list(v) if v is not None else [0] * k
        self.ncd, self.ngd = z(node_c, self.N), z(node_g, self.N)
        self.ecd, self.egd = z(edge_c, self.E), z(edge_g, self.E)
        E, N = self.E, self.N
        off = {t: [0] * N for t in _TABLES}
        xo = 0
        for i in range(N):
            off["x_state"][i] = xo
            if i < E:
                xo += self.sd[i]
                off["x_control"][i] = xo
                xo += self.cd[i]
        self.x_dim = self.sd[E] + sum(self.sd[e] + self.cd[e] for e in range(E))
        yo = 0
        for i in range(N):
            off["y_dyn"][i] = yo
            yo += self.sd[i]
            off["y_node_c"][i] = yo
            yo += self.ncd[i]
        for e in range(E):
            off["y_edge_c"][e] = yo
            yo += self.ecd[e]
        zo = 0
        for i in range(N):
            off["z_node"][i] = zo
            zo += self.ngd[i]
        for e in range(E):
            off["z_edge"][e] = zo
            zo += self.egd[e]
        self.y_dim, self.z_dim = yo, zo
        self.kkt_dim = self.x_dim + yo + zo
        self.off = off
        # model arena: node i then edge i
        self.node_off = {b: [0] * N for b in NODE_BLOCKS}
        self.edge_off = {b: [0] * E for b in EDGE_BLOCKS}
        at = 0
        for i in range(N):
            for b, sz in zip(NODE_BLOCKS, self.node_shapes(i)):
                self.node_off[b][i] = at
                at += sz[0] * sz[1]
            if i < E:
                for b, sz in zip(EDGE_BLOCKS, self.edge_shapes(i)):
                    self.edge_off[b][i] = at
                    at += sz[0] * sz[1]
        self.model_len = at
        # theta arena: node i then edge i (kkt_oracle.h)
        self.theta_node_off = {b: [0] * N for b in THETA_NODE_BLOCKS}
        self.theta_edge_off = {b: [0] * E for b in THETA_EDGE_BLOCKS}
        at = 0
        if self.p > 0:
            for i in range(N):
                for b, sz in zip(THETA_NODE_BLOCKS, self.theta_node_shapes(i)):
                    self.theta_node_off[b][i] = at
                    at += sz[0] * sz[1]
                if i < E:
                    for b, sz in zip(THETA_EDGE_BLOCKS, self.theta_edge_shapes(i)):
                        self.theta_edge_off[b][i] = at
                        at += sz[0] * sz[1]
        self.theta_len = at
        self.full_dim = self.kkt_dim + self.p  # [x | theta | y | z]

    def theta_node_shapes(self, i):
        p = self.p
        return [(self.sd[i], p), (self.ncd[i], p), (self.ngd[i], p), (p, p)]

    def theta_edge_shapes(self, e):
        p = self.p
        return [(self.sd[self.parents[e]], p), (self.cd[e], p), (self.sd[self.children[e]], p), (self.ecd[e], p),
                (self.egd[e], p), (p, p)]

    def pack_theta(self, nodes, edges):
        out = np.zeros(self.theta_len)
        for i in range(self.N):
            for b, shp in zip(THETA_NODE_BLOCKS, self.theta_node_shapes(i)):
                a = np.asarray(nodes[i][b], dtype=np.float64).reshape(shp)
                out[self.theta_node_off[b][i]:self.theta_node_off[b][i] + a.size] = a.reshape(-1, order="F")
        for e in range(self.E):
            for b, shp in zip(THETA_EDGE_BLOCKS, self.theta_edge_shapes(e)):
                a = np.asarray(edges[e][b], dtype=np.float64).reshape(shp)
                out[self.theta_edge_off[b][e]:self.theta_edge_off[b][e] + a.size] = a.reshape(-1, order="F")
        return out

    def unpack_theta(self, arena):
        arena = np.asarray(arena, dtype=np.float64)
        nodes = [{b: arena[self.theta_node_off[b][i]:self.theta_node_off[b][i] + s[0] * s[1]].reshape(s, order="F")
                  for b, s in zip(THETA_NODE_BLOCKS, self.theta_node_shapes(i))} for i in range(self.N)]
        edges = [{b: arena[self.theta_edge_off[b][e]:self.theta_edge_off[b][e] + s[0] * s[1]].reshape(s, order="F")
                  for b, s in zip(THETA_EDGE_BLOCKS, self.theta_edge_shapes(e))} for e in range(self.E)]
        return nodes, edges

    def node_shapes(self, i):
        n = self.sd[i]
        return [(n, n), (self.ncd[i], n), (self.ngd[i], n)]

    def edge_shapes(self, e):
        np_, nc, m = self.sd[self.parents[e]], self.sd[self.children[e]], self.cd[e]
        c, g = self.ecd[e], self.egd[e]
        return [(np_, np_), (np_, m), (m, m), (nc, np_), (nc, m), (c, np_), (c, m), (g, np_), (g, m)]

    def pack_model(self, nodes, edges):
        """nodes[i][name] / edges[e][name]: numpy (row, col) arrays -> flat arena."""
        out = np.zeros(self.model_len)
        for i in range(self.N):
            for b, shp in zip(NODE_BLOCKS, self.node_shapes(i)):
                a = np.asarray(nodes[i][b], dtype=np.float64).reshape(shp)
                out[self.node_off[b][i]:self.node_off[b][i] + a.size] = a.reshape(-1, order="F")
        for e in range(self.E):
            for b, shp in zip(EDGE_BLOCKS, self.edge_shapes(e)):
                a = np.asarray(edges[e][b], dtype=np.float64).reshape(shp)
                out[self.edge_off[b][e]:self.edge_off[b][e] + a.size] = a.reshape(-1, order="F")
        return out

    def unpack_model(self, model):
        model = np.asarray(model, dtype=np.float64)
        nodes, edges = [], []
        for i in range(self.N):
            nodes.append({b: model[self.node_off[b][i]:self.node_off[b][i] + s[0] * s[1]].reshape(s, order="F")
                          for b, s in zip(NODE_BLOCKS, self.node_shapes(i))})
        for e in range(self.E):
            edges.append({b: model[self.edge_off[b][e]:self.edge_off[b][e] + s[0] * s[1]].reshape(s, order="F")
                          for b, s in zip(EDGE_BLOCKS, self.edge_shapes(e))})
        return nodes, edges


def dense_kkt_blocks(dims, model, theta_model=None):
    """The blocks H, C, G of the KKT matrix (x-space columns: [stagewise x | theta])."""
    return dense_kkt_matrix(dims, model, None, None, None, None, theta_model, blocks_only=True)


def dense_kkt_matrix(dims, model, w, r1, r2, r3, theta_model=None, blocks_only=False):
    """Full regularized KKT matrix from the model blocks (independent of the C code)."""
    nodes, edges = dims.unpack_model(model)
    xd, yd, zd = dims.x_dim, dims.y_dim, dims.z_dim
    H = np.zeros((xd, xd))
    C = np.zeros((yd, xd))
    G = np.zeros((zd, xd))
    o = dims.off
    sl = lambda start, size: slice(start, start + size)
    for i in range(dims.N):
        n = dims.sd[i]
        xs = sl(o["x_state"][i], n)
        H[xs, xs] += nodes[i]["d2L_dx2"]
        C[sl(o["y_node_c"][i], dims.ncd[i]), xs] += nodes[i]["dc_dx"]
        G[sl(o["z_node"][i], dims.ngd[i]), xs] += nodes[i]["dg_dx"]
    root = dims.root
    C[sl(o["y_dyn"][root], dims.sd[root]), sl(o["x_state"][root], dims.sd[root])] -= np.eye(dims.sd[root])
    for e in range(dims.E):
        p, ch, m = dims.parents[e], dims.children[e], dims.cd[e]
        xp, xc, xu = sl(o["x_state"][p], dims.sd[p]), sl(o["x_state"][ch], dims.sd[ch]), sl(o["x_control"][e], m)
        ed = edges[e]
        H[xp, xp] += ed["d2L_dx2"]
        H[xp, xu] += ed["d2L_dxdu"]
        H[xu, xp] += ed["d2L_dxdu"].T
        H[xu, xu] += ed["d2L_du2"]
        yd_rows = sl(o["y_dyn"][ch], dims.sd[ch])
        C[yd_rows, xp] += ed["ddyn_dx"]
        C[yd_rows, xu] += ed["ddyn_du"]
        C[yd_rows, xc] -= np.eye(dims.sd[ch])
        yc = sl(o["y_edge_c"][e], dims.ecd[e])
        C[yc, xp] += ed["dc_dx"]
        C[yc, xu] += ed["dc_du"]
        zg = sl(o["z_edge"][e], dims.egd[e])
        G[zg, xp] += ed["dg_dx"]
        G[zg, xu] += ed["dg_du"]
    if theta_model is not None and dims.p > 0:
        # x = [stagewise x | theta]: widen H, C, G by the theta columns (helpers.cpp:190-240)
        p = dims.p
        tn, te = dims.unpack_theta(theta_model)
        Hf = np.zeros((xd + p, xd + p))
        Hf[:xd, :xd] = H
        Cf = np.hstack([C, np.zeros((yd, p))])
        Gf = np.hstack([G, np.zeros((zd, p))])
        th = slice(xd, xd + p)
        for i in range(dims.N):
            xs = sl(o["x_state"][i], dims.sd[i])
            Hf[xs, th] += tn[i]["d2L_dxdtheta"]
            Hf[th, xs] += tn[i]["d2L_dxdtheta"].T
            Hf[th, th] += tn[i]["d2L_dtheta2"]
            Cf[sl(o["y_node_c"][i], dims.ncd[i]), th] += tn[i]["dc_dtheta"]
            Gf[sl(o["z_node"][i], dims.ngd[i]), th] += tn[i]["dg_dtheta"]
        for e in range(dims.E):
            pa, ch = dims.parents[e], dims.children[e]
            xp, xu = sl(o["x_state"][pa], dims.sd[pa]), sl(o["x_control"][e], dims.cd[e])
            Hf[xp, th] += te[e]["d2L_dxdtheta"]
            Hf[th, xp] += te[e]["d2L_dxdtheta"].T
            Hf[xu, th] += te[e]["d2L_dudtheta"]
            Hf[th, xu] += te[e]["d2L_dudtheta"].T
            Hf[th, th] += te[e]["d2L_dtheta2"]
            Cf[sl(o["y_dyn"][ch], dims.sd[ch]), th] += te[e]["ddyn_dtheta"]
            Cf[sl(o["y_edge_c"][e], dims.ecd[e]), th] += te[e]["dc_dtheta"]
            Gf[sl(o["z_edge"][e], dims.egd[e]), th] += te[e]["dg_dtheta"]
        H, C, G, xd = Hf, Cf, Gf, xd + p
    if blocks_only:
        return H, C, G
    K = np.zeros((xd + yd + zd, xd + yd + zd))
    K[:xd, :xd] = H + np.diag(r1)
    K[:xd, xd:xd + yd] = C.T
    K[:xd, xd + yd:] = G.T
    K[xd:xd + yd, :xd] = C
    K[xd:xd + yd, xd:xd + yd] = -np.diag(r2)
    K[xd + yd:, :xd] = G
    K[xd + yd:, xd + yd:] = -np.diag(np.asarray(w) + np.asarray(r3))
    return K


class KKTOracle:
    def __init__(self, dims, null_dims=False):
        self.dims = dims
        L = _lib()
        args = [_ints(dims.parents), _ints(dims.children), _ints(dims.sd), _ints(dims.cd)]
        args += [None] * 4 if null_dims else [_ints(dims.ncd), _ints(dims.ngd), _ints(dims.ecd), _ints(dims.egd)]
        self._keep = args
        self.h = L.kkt_oracle_create(dims.E, dims.root, *args)
        if dims.p > 0:
            L.kkt_oracle_set_theta(self.h, dims.p)
            assert L.kkt_oracle_theta_len(self.h) == dims.theta_len

    def theta_offset(self, block, index):
        return _lib().kkt_oracle_theta_offset(self.h, block, index)

    def factor_theta(self, model, theta_model, w, r1, r2, r3):
        keep = [_f64(a) for a in (model, theta_model, w, r1, r2, r3)]
        return _lib().kkt_oracle_factor_theta(self.h, *[k[1] for k in keep])

    def solve_theta(self, model, theta_model, b):
        keep = [_f64(a) for a in (model, theta_model, b)]
        sol = np.zeros(self.dims.full_dim)
        _lib().kkt_oracle_solve_theta(self.h, *[k[1] for k in keep], sol.ctypes.data)
        return sol

    def add_Kx_to_y_theta(self, model, theta_model, w, r1, r2, r3, x):
        keep = [_f64(a) for a in (model, theta_model, w, r1, r2, r3, x)]
        out = np.zeros(self.dims.full_dim)
        _lib().kkt_oracle_add_Kx_to_y_theta(self.h, *[k[1] for k in keep], out.ctypes.data)
        return out

    def dim(self, which):
        return _lib().kkt_oracle_dim(self.h, which)

    def model_offset(self, block, index):
        return _lib().kkt_oracle_model_offset(self.h, block, index)

    def vector_offset(self, table, index):
        return _lib().kkt_oracle_vector_offset(self.h, table, index)

    def factor(self, model, w, r1, r2, r3):
        keep = [_f64(a) for a in (model, w, r1, r2, r3)]
        self._model = keep[0]
        return _lib().kkt_oracle_factor(self.h, *[k[1] for k in keep])

    def solve(self, model, b):
        (m, mp), (bb, bp) = _f64(model), _f64(b)
        sol = np.zeros(self.dims.kkt_dim)
        _lib().kkt_oracle_solve(self.h, mp, bp, sol.ctypes.data)
        return sol

    def add_Kx_to_y(self, model, w, r1, r2, r3, x, y=None):
        keep = [_f64(a) for a in (model, w, r1, r2, r3, x)]
        out = np.zeros(self.dims.kkt_dim) if y is None else np.array(y, dtype=np.float64)
        _lib().kkt_oracle_add_Kx_to_y(self.h, *[k[1] for k in keep], out.ctypes.data)
        return out

    # the five block operators (helpers.hpp:20-24): (input space, output space) of each
    BLOCK_SPACES = {"Hx": ("x", "x"), "Cx": ("x", "y"), "CTx": ("y", "x"), "Gx": ("x", "z"), "GTx": ("z", "x")}

    def space_dim(self, space):
        d = self.dims
        return {"x": d.x_dim + d.p, "y": d.y_dim, "z": d.z_dim}[space]

    def add_block_to_y(self, name, model, x, y=None, theta_model=None):
        """y += (block) x for name in BLOCK_SPACES; x-space vectors are [stagewise x | theta]."""
        src, dst = self.BLOCK_SPACES[name]
        keep = [_f64(model), _f64(x)]
        assert keep[1][0].size == self.space_dim(src)
        tm = _f64(theta_model) if theta_model is not None else (None, None)
        out = np.zeros(self.space_dim(dst)) if y is None else np.array(y, dtype=np.float64)
        getattr(_lib(), f"kkt_oracle_add_{name}_to_y")(self.h, keep[0][1], tm[1], keep[1][1], out.ctypes.data)
        return out

    def lqr_block(self, name, index, size):
        ptr = _lib().kkt_oracle_lqr_block(self.h, name.encode(), index)
        return np.array([ptr[i] for i in range(size)])

    def batch(self, model, w, r1, r2, r3, b, threads=1):
        keep = [_f64(a) for a in (model, w, r1, r2, r3, b)]
        batch = keep[0][0].shape[0]
        sol = np.zeros((batch, self.dims.kkt_dim))
        status = np.zeros(batch, dtype=np.int32)
        _lib().kkt_oracle_batch(self.h, batch, *[k[1] for k in keep], sol.ctypes.data, status.ctypes.data,
                                threads)
        return sol, status

    def close(self):
        if self.h:
            _lib().kkt_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
