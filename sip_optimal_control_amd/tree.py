"""General tree / variable-dimension LQR on the GPU (plumbing over the C ABI).

Mirrors the reference's `LQR` life cycle (lqr.hpp:189-194) for `batch`
instances of one topology: construct (topology compiled and latched), then
`factor()` -> statuses, `solve()` -> x, u, y, any number of times.
"""
import ctypes

import numpy as np
import torch

from ._lib import LQRLibraryError, load_library, resolve_device
from .chain import _check

_NODE_IN = ("Q", "q", "c", "delta")
_EDGE_IN = ("A", "B", "M", "R", "r")


def _ints(values):
    return (ctypes.c_int * max(1, len(values)))(*[int(v) for v in values])


def compile_topology(num_edges, root, parents, children):
    """Product-side host restatement of compile_topology_data (lqr.cpp:563-631).
    Returns (status, dict of arrays)."""
    lib = load_library()
    E, N = num_edges, num_edges + 1
    co, ce, po, cho = _ints([0] * (N + 1)), _ints([0] * E), _ints([0] * E), _ints([0] * E)
    pre, post, marks = _ints([0] * N), _ints([0] * N), _ints([0] * N)
    st = lib.sip_lqr_compile_topology(E, root, _ints(parents) if parents is not None else None,
                                      _ints(children) if children is not None else None,
                                      co, ce, po, cho, pre, post, marks)
    return st, {"child_offsets": list(co)[:N + 1], "child_edges": list(ce)[:E],
                "preorder_nodes": list(pre)[:N], "postorder_nodes": list(post)[:N]}


class BatchedTreeLQR:
    def __init__(self, parents, children, state_dims, control_dims, batch=1, root=0, device="cuda:0"):
        self._lib = load_library()
        self.E = len(control_dims)
        self.N = self.E + 1
        self.parents, self.children = list(parents), list(children)
        self.state_dims, self.control_dims = list(state_dims), list(control_dims)
        self.batch = int(batch)
        self.device = resolve_device(device)  # explicit ordinal; raises without a HIP device
        h = ctypes.c_void_p()
        _check(self._lib.sip_lqr_tree_plan_create(self.batch, self.E, root, _ints(self.parents),
                                                  _ints(self.children), _ints(self.state_dims),
                                                  _ints(self.control_dims), self.device.index,
                                                  ctypes.byref(h)), "sip_lqr_tree_plan_create")
        self._plan = h
        self.topology_status = self._lib.sip_lqr_tree_topology_status(h)
        self.in_len = self._lib.sip_lqr_tree_input_len(h)
        self.ws_len = self._lib.sip_lqr_tree_work_len(h)
        self.out_len = self._lib.sip_lqr_tree_output_len(h)
        f64 = torch.float64
        self.input = torch.zeros(self.batch, max(1, self.in_len), dtype=f64, device=self.device)
        self.work = torch.zeros(self.batch, max(1, self.ws_len), dtype=f64, device=self.device)
        self.output = torch.zeros(self.batch, max(1, self.out_len), dtype=f64, device=self.device)
        self.status = torch.full((self.batch,), -1, dtype=torch.int32, device=self.device)

    def offset(self, arena, kind, index):
        off = self._lib.sip_lqr_tree_offset(self._plan, arena, kind, index)
        if off == ctypes.c_size_t(-1).value:
            raise IndexError((arena, kind, index))
        return off

    def pack(self, problems):
        """problems: list (len batch) of blocks dicts (numpy (row, col) arrays) -> device input arena."""
        host = np.zeros((self.batch, max(1, self.in_len)))
        for b, blocks in enumerate(problems):
            for node in range(self.N):
                o = self.offset(0, 0, node)
                for name in _NODE_IN:
                    a = np.asarray(blocks[name][node], dtype=np.float64).reshape(-1, order="F")
                    host[b, o:o + a.size] = a
                    o += a.size
            for e in range(self.E):
                o = self.offset(0, 1, e)
                for name in _EDGE_IN:
                    a = np.asarray(blocks[name][e], dtype=np.float64).reshape(-1, order="F")
                    host[b, o:o + a.size] = a
                    o += a.size
        self.input.copy_(torch.from_numpy(host))

    def factor(self):
        s = torch.cuda.current_stream(self.device)
        _check(self._lib.sip_lqr_tree_factor(self._plan, ctypes.c_void_p(self.input.data_ptr()),
                                             ctypes.c_void_p(self.work.data_ptr()),
                                             ctypes.c_void_p(self.status.data_ptr()),
                                             ctypes.c_void_p(s.cuda_stream)), "sip_lqr_tree_factor")
        return self.status

    @property
    def kernel_name(self):
        return self._lib.sip_lqr_tree_kernel_name(self._plan).decode()

    def factor_solve(self, workspace=False):
        """factor_with_status() + solve() as one fused sweep (size-class kernel of csrc/tree_qw16.hpp
        where the tree fits one, the general engine otherwise): output, status, K and k of `work`;
        workspace=True: every LQR::Workspace field of `work` (sip_lqr_tree_factor_solve_workspace)."""
        s = torch.cuda.current_stream(self.device)
        entry = self._lib.sip_lqr_tree_factor_solve_workspace if workspace else self._lib.sip_lqr_tree_factor_solve
        need = self._lib.sip_lqr_tree_fused_scratch_bytes(self._plan)
        if getattr(self, "_scratch", None) is None or self._scratch.numel() < need:
            self._scratch = torch.empty(max(1, need), dtype=torch.uint8, device=self.device)
        _check(entry(self._plan, ctypes.c_void_p(self.input.data_ptr()),
                     ctypes.c_void_p(self.work.data_ptr()),
                     ctypes.c_void_p(self.output.data_ptr()),
                     ctypes.c_void_p(self.status.data_ptr()),
                     ctypes.c_void_p(self._scratch.data_ptr()),
                     ctypes.c_void_p(s.cuda_stream)), "sip_lqr_tree_factor_solve")
        return self.output, self.status

    def solve(self):
        s = torch.cuda.current_stream(self.device)
        _check(self._lib.sip_lqr_tree_solve(self._plan, ctypes.c_void_p(self.input.data_ptr()),
                                            ctypes.c_void_p(self.work.data_ptr()),
                                            ctypes.c_void_p(self.output.data_ptr()),
                                            ctypes.c_void_p(self.status.data_ptr()),
                                            ctypes.c_void_p(s.cuda_stream)), "sip_lqr_tree_solve")
        return self.output

    def unpack_solution(self, b=0):
        out = self.output[b].cpu().numpy()
        x, y, u = [], [], []
        for node, n in enumerate(self.state_dims):
            o = self.offset(2, 0, node)
            x.append(out[o:o + n].copy())
            y.append(out[o + n:o + 2 * n].copy())
        for e, m in enumerate(self.control_dims):
            o = self.offset(2, 1, e)
            u.append(out[o:o + m].copy())
        return x, u, y

    def unpack_gains(self, b=0):
        ws = self.work[b].cpu().numpy()
        max_n = max(self.state_dims) if self.state_dims else 0
        Ks, ks = [], []
        for e, m in enumerate(self.control_dims):
            n = self.state_dims[self.parents[e]]
            o = self.offset(1, 1, e) + max_n * max_n
            Ks.append(ws[o:o + m * n].reshape((m, n), order="F").copy())
            o += m * n + m * m
            ks.append(ws[o:o + m].copy())
        return Ks, ks

    def topology_arrays(self):
        names = ["child_offsets", "child_edges", "preorder_nodes", "postorder_nodes"]
        sizes = [self.N + 1, self.E, self.N, self.N]
        out = {}
        for which, (name, size) in enumerate(zip(names, sizes)):
            ptr = self._lib.sip_lqr_tree_topology_array(self._plan, which)
            out[name] = [ptr[i] for i in range(size)]
        return out

    def close(self):
        if getattr(self, "_plan", None):
            self._lib.sip_lqr_tree_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
