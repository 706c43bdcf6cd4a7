"""ctypes front-end of the CPU parity oracle (oracle/lqr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblqr_oracle.so")
_SO_NATIVE = os.path.join(_HERE, "liblqr_oracle_native.so")
_lib = None
_lib_native = None

STATUS_NAMES = {0: "SUCCESS", 1: "INVALID_DELTA", 2: "F_FACTORIZATION_FAILURE",
                3: "G_FACTORIZATION_FAILURE", 4: "INVALID_TOPOLOGY"}

_DPP = ctypes.POINTER(ctypes.POINTER(ctypes.c_double))
_IP = ctypes.POINTER(ctypes.c_int)


class _Problem(ctypes.Structure):
    _fields_ = [("num_edges", ctypes.c_int), ("root", ctypes.c_int),
                ("edge_parents", _IP), ("edge_children", _IP),
                ("state_dims", _IP), ("control_dims", _IP)] + \
               [(name, _DPP) for name in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")]


class _Workspace(ctypes.Structure):
    _fields_ = [(name, _DPP) for name in ("W", "K", "V", "G_factor", "F_factor", "sqrt_delta",
                                          "sqrt_delta_inv", "k", "v")] + \
               [(name, ctypes.POINTER(ctypes.c_double)) for name in ("G", "g", "H", "h", "F", "f")] + \
               [(name, _IP) for name in ("child_offsets", "child_edges", "edge_parents",
                                         "edge_children", "preorder_nodes", "postorder_nodes",
                                         "node_marks")] + \
               [("traversal_status", ctypes.c_int), ("num_edges", ctypes.c_int)]


def build(force=False):
    sources = [os.path.join(_HERE, f) for f in ("lqr_oracle.c", "lqr_oracle.h", "kkt_oracle.c", "kkt_oracle.h")]
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in sources):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_native(force=False):
    """Second build of the same two C files for the TIMED cpu_baseline of bench.py only:
    -O3 -march=native with FMA contraction on, built on the host that times it
    (the sidecar records the CPU model: a library that travelled from another machine is rebuilt).
    The parity oracle stays the -ffp-contract=off build above."""
    sources = [os.path.join(_HERE, f) for f in ("lqr_oracle.c", "lqr_oracle.h", "kkt_oracle.c", "kkt_oracle.h")]
    side = _SO_NATIVE + ".host"
    tag = _cpu_model()
    fresh = os.path.exists(_SO_NATIVE) and os.path.exists(side) and open(side).read().strip() == tag and \
        os.path.getmtime(_SO_NATIVE) >= max(os.path.getmtime(f) for f in sources)
    if force or not fresh:
        if os.path.exists(_SO_NATIVE):
            os.remove(_SO_NATIVE)
        subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
        with open(side, "w") as f:
            f.write(tag + "\n")
    return _SO_NATIVE


def lib(native=False):
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            _lib_native = _bind(ctypes.CDLL(build_native()))
        return _lib_native
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _bind(ctypes.CDLL(_SO))
    return _lib


def _bind(L):
    if True:
        L.lqr_oracle_workspace_reserve.argtypes = [ctypes.POINTER(_Workspace), ctypes.POINTER(_Problem)]
        L.lqr_oracle_workspace_free.argtypes = [ctypes.POINTER(_Workspace)]
        L.lqr_oracle_workspace_free.restype = None
        L.lqr_oracle_compile_topology.argtypes = [ctypes.POINTER(_Problem), ctypes.POINTER(_Workspace)]
        L.lqr_oracle_factor.argtypes = [ctypes.POINTER(_Problem), ctypes.POINTER(_Workspace)]
        L.lqr_oracle_solve.argtypes = [ctypes.POINTER(_Problem), ctypes.POINTER(_Workspace), _DPP, _DPP, _DPP]
        L.lqr_oracle_solve.restype = None
        L.lqr_oracle_chain_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long] + \
            [ctypes.c_void_p] * 5 + [ctypes.c_int]
        for name in ("mats", "vecs", "gains"):
            fn = getattr(L, f"lqr_oracle_chain_{name}_len")
            fn.argtypes = [ctypes.c_int] * 3
            fn.restype = ctypes.c_long
    return L


def _ptr_table(arrays):
    tab = (ctypes.POINTER(ctypes.c_double) * max(1, len(arrays)))()
    for i, a in enumerate(arrays):
        tab[i] = a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    return tab


def _int_array(values):
    return (ctypes.c_int * max(1, len(values)))(*values)


class TreeLQR:
    """One reference-shaped problem: mirrors `LQR` (lqr.hpp:66-200) on the oracle.

    blocks: dict name -> list of numpy float64 arrays (matrices in numpy
    (row, col) convention; stored column-major for the oracle).
    """

    def __init__(self, parents, children, state_dims, control_dims, blocks, root=0,
                 null_topology=False):
        self.E = len(control_dims)
        self.state_dims = list(state_dims)
        self.control_dims = list(control_dims)
        self._keep = []
        self._parents = _int_array(list(parents))
        self._children = _int_array(list(children))
        self._sd = _int_array(self.state_dims)
        self._cd = _int_array(self.control_dims)
        prob = _Problem()
        prob.num_edges = self.E
        prob.root = root
        if not null_topology:
            prob.edge_parents = self._parents
            prob.edge_children = self._children
        prob.state_dims = self._sd
        prob.control_dims = self._cd
        for name in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta"):
            arrs = [np.asfortranarray(np.asarray(b, dtype=np.float64)) for b in blocks[name]]
            flat = [np.ascontiguousarray(a.reshape(-1, order="F")) for a in arrs]
            self._keep.append(flat)
            tab = _ptr_table(flat)
            self._keep.append(tab)
            setattr(prob, name, tab)
        self.prob = prob
        self.ws = _Workspace()
        lib().lqr_oracle_workspace_reserve(ctypes.byref(self.ws), ctypes.byref(self.prob))
        self.topology_status = lib().lqr_oracle_compile_topology(ctypes.byref(self.prob),
                                                                 ctypes.byref(self.ws))

    def factor(self):
        return lib().lqr_oracle_factor(ctypes.byref(self.prob), ctypes.byref(self.ws))

    def solve(self):
        N = self.E + 1
        x = [np.zeros(d) for d in self.state_dims]
        y = [np.zeros(d) for d in self.state_dims]
        u = [np.zeros(d) for d in self.control_dims]
        tx, tu, ty = _ptr_table(x), _ptr_table(u), _ptr_table(y)
        lib().lqr_oracle_solve(ctypes.byref(self.prob), ctypes.byref(self.ws), tx, tu, ty)
        assert len(x) == N
        return x, u, y

    def topology_arrays(self):
        N = self.E + 1
        ws = self.ws
        return {"child_offsets": [ws.child_offsets[i] for i in range(N + 1)],
                "child_edges": [ws.child_edges[i] for i in range(self.E)],
                "preorder_nodes": [ws.preorder_nodes[i] for i in range(N)],
                "postorder_nodes": [ws.postorder_nodes[i] for i in range(N)]}

    def gains(self):
        out_K, out_k = [], []
        for e in range(self.E):
            m = self.control_dims[e]
            n = self.state_dims[self.ws.edge_parents[e]]
            K = np.array([self.ws.K[e][i] for i in range(m * n)]).reshape((m, n), order="F")
            out_K.append(K)
            out_k.append(np.array([self.ws.k[e][i] for i in range(m)]))
        return out_K, out_k

    def close(self):
        if self.ws is not None:
            lib().lqr_oracle_workspace_free(ctypes.byref(self.ws))
            self.ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chain_batch(n, m, T, mats, vecs, threads=1, want_gains=True, native=False):
    """Run the oracle over a packed uniform-chain batch (numpy float64 [batch, len]).
    native: the -march=native / FMA build (timing only, never parity)."""
    L = lib(native)
    mats = np.ascontiguousarray(mats, dtype=np.float64)
    vecs = np.ascontiguousarray(vecs, dtype=np.float64)
    batch = mats.shape[0]
    assert mats.shape[1] == L.lqr_oracle_chain_mats_len(n, m, T)
    assert vecs.shape == (batch, L.lqr_oracle_chain_vecs_len(n, m, T))
    sol = np.zeros_like(vecs)
    gains = np.zeros((batch, L.lqr_oracle_chain_gains_len(n, m, T))) if want_gains else None
    status = np.zeros(batch, dtype=np.int32)
    L.lqr_oracle_chain_batch(n, m, T, batch, mats.ctypes.data, vecs.ctypes.data, sol.ctypes.data,
                             gains.ctypes.data if want_gains else None, status.ctypes.data, threads)
    return sol, gains, status
