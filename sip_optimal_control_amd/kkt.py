"""Batched Newton-KKT step on the GPU (plumbing over include/sip_kkt_amd.h).

Mirrors the reference's `CallbackProvider` (helpers.hpp:7-33) for `batch`
problems of one topology / dimension table, theta_dim == 0:
`factor(w, r1, r2, r3)` -> statuses (0 where the reference returns true),
`solve(b)` -> sol, `add_Kx_to_y(...)`; plus the fused `factor_solve`.
All arrays are device tensors [batch, len] in the reference's flattened
variable ordering (types.cpp:24-64); `model` is the flat arena of model
callback outputs documented in the header.
"""
import ctypes

import torch

from ._lib import LQRLibraryError, load_library, resolve_device
from .chain import _check

NODE_BLOCKS = ("d2L_dx2", "dc_dx", "dg_dx")
EDGE_BLOCKS = ("d2L_dx2", "d2L_dxdu", "d2L_du2", "ddyn_dx", "ddyn_du", "dc_dx", "dc_du", "dg_dx", "dg_du")
VECTOR_TABLES = ("x_state", "x_control", "y_dyn", "y_node_c", "y_edge_c", "z_node", "z_edge")
STATUS_NONPOSITIVE_REGULARIZATION = 5
STATUS_INVALID_INPUT = 6


def _ints(values):
    if values is None:
        return None
    return (ctypes.c_int * max(1, len(values)))(*[int(v) for v in values])


class BatchedNewtonKKT:
    def __init__(self, parents, children, state_dims, control_dims, node_c_dims=None, node_g_dims=None,
                 edge_c_dims=None, edge_g_dims=None, batch=1, root=0, device="cuda:0", theta_dim=0):
        self._lib = load_library()
        self.device = resolve_device(device)  # explicit ordinal; raises without a HIP device
        self.E, self.N, self.batch = len(control_dims), len(control_dims) + 1, int(batch)
        h = ctypes.c_void_p()
        _check(self._lib.sip_kkt_plan_create(self.batch, self.E, root, _ints(parents), _ints(children),
                                             _ints(state_dims), _ints(control_dims), _ints(node_c_dims),
                                             _ints(node_g_dims), _ints(edge_c_dims), _ints(edge_g_dims),
                                             self.device.index, ctypes.byref(h)), "sip_kkt_plan_create")
        self._plan = h
        self.input_status = self._lib.sip_kkt_input_status(h)
        self.x_dim, self.y_dim, self.z_dim, self.model_len = (self._lib.sip_kkt_len(h, k) for k in range(4))
        self.kkt_dim = self.x_dim + self.y_dim + self.z_dim
        self.kernel_name = self._lib.sip_kkt_kernel_name(h).decode()
        self.work = torch.empty(max(1, self._lib.sip_kkt_work_bytes(h)), dtype=torch.uint8, device=self.device)
        self.theta_dim = int(theta_dim)
        self.theta_len = 0
        if self.theta_dim > 0:  # global variables: x = [stagewise x | theta]
            _check(self._lib.sip_kkt_plan_set_theta(h, self.theta_dim), "sip_kkt_plan_set_theta")
            self.theta_len = self._lib.sip_kkt_theta_len(h)
            self.kernel_name = self._lib.sip_kkt_kernel_name(h).decode()  # (says which theta passes the plan runs)
            self.theta_work = torch.empty(max(1, self._lib.sip_kkt_theta_work_bytes(h)), dtype=torch.uint8,
                                          device=self.device)
        self.full_dim = self.kkt_dim + self.theta_dim
        self.status = torch.full((self.batch,), -1, dtype=torch.int32, device=self.device)

    def model_offset(self, block, index):
        off = self._lib.sip_kkt_model_offset(self._plan, block, index)
        if off == ctypes.c_size_t(-1).value:
            raise IndexError((block, index))
        return off

    def vector_offset(self, table, index):
        off = self._lib.sip_kkt_vector_offset(self._plan, table, index)
        if off == ctypes.c_size_t(-1).value:
            raise IndexError((table, index))
        return off

    def _ptr(self, t, length):
        if t is None:
            return None
        if t.dtype != torch.float64 or not t.is_contiguous() or t.device != self.device or \
                t.numel() != self.batch * length:
            raise ValueError(f"expected a contiguous float64 [{self.batch}, {length}] tensor on {self.device}")
        return t.data_ptr()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def factor(self, model, w, r1, r2, r3):
        _check(self._lib.sip_kkt_factor(self._plan, self._ptr(model, self.model_len), self._ptr(w, self.z_dim),
                                        self._ptr(r1, self.x_dim), self._ptr(r2, self.y_dim),
                                        self._ptr(r3, self.z_dim), self.work.data_ptr(), self.status.data_ptr(),
                                        self._stream()), "sip_kkt_factor")
        return self.status

    def solve(self, model, b, sol=None):
        if sol is None:
            sol = torch.zeros(self.batch, self.kkt_dim, dtype=torch.float64, device=self.device)
        _check(self._lib.sip_kkt_solve(self._plan, self._ptr(model, self.model_len), self._ptr(b, self.kkt_dim),
                                       self._ptr(sol, self.kkt_dim), self.work.data_ptr(), self.status.data_ptr(),
                                       self._stream()), "sip_kkt_solve")
        return sol

    def factor_solve(self, model, w, r1, r2, r3, b, sol=None):
        if sol is None:
            sol = torch.zeros(self.batch, self.kkt_dim, dtype=torch.float64, device=self.device)
        _check(self._lib.sip_kkt_factor_solve(self._plan, self._ptr(model, self.model_len),
                                              self._ptr(w, self.z_dim), self._ptr(r1, self.x_dim),
                                              self._ptr(r2, self.y_dim), self._ptr(r3, self.z_dim),
                                              self._ptr(b, self.kkt_dim), self._ptr(sol, self.kkt_dim),
                                              self.work.data_ptr(), self.status.data_ptr(), self._stream()),
               "sip_kkt_factor_solve")
        return sol, self.status

    def add_Kx_to_y(self, model, w, r1, r2, r3, x, y=None):
        if y is None:
            y = torch.zeros(self.batch, self.kkt_dim, dtype=torch.float64, device=self.device)
        _check(self._lib.sip_kkt_add_Kx_to_y(self._plan, self._ptr(model, self.model_len), self._ptr(w, self.z_dim),
                                             self._ptr(r1, self.x_dim), self._ptr(r2, self.y_dim),
                                             self._ptr(r3, self.z_dim), self._ptr(x, self.kkt_dim),
                                             self._ptr(y, self.kkt_dim), self._stream()), "sip_kkt_add_Kx_to_y")
        return y

    # ---- the five block operators (CallbackProvider::add_{H,C,CT,G,GT}x_to_y, helpers.hpp:20-24) ----
    BLOCK_SPACES = {"Hx": ("x", "x"), "Cx": ("x", "y"), "CTx": ("y", "x"), "Gx": ("x", "z"), "GTx": ("z", "x")}

    def space_dim(self, space, theta=False):
        return {"x": self.x_dim + (self.theta_dim if theta else 0), "y": self.y_dim, "z": self.z_dim}[space]

    def add_block_to_y(self, name, model, x, y=None, theta_model=None):
        """y += (block) x for name in BLOCK_SPACES; vectors are [batch, len of their space]; with
        theta_model the x-space vectors are [stagewise x | theta]."""
        src, dst = self.BLOCK_SPACES[name]
        theta = theta_model is not None
        if y is None:
            y = torch.zeros(self.batch, self.space_dim(dst, theta), dtype=torch.float64, device=self.device)
        xs, ys = self._ptr(x, self.space_dim(src, theta)), self._ptr(y, self.space_dim(dst, theta))
        if theta:
            fn = getattr(self._lib, f"sip_kkt_add_{name}_to_y_theta")
            _check(fn(self._plan, self._ptr(model, self.model_len), self._ptr(theta_model, self.theta_len), xs, ys,
                      self._stream()), f"sip_kkt_add_{name}_to_y_theta")
        else:
            fn = getattr(self._lib, f"sip_kkt_add_{name}_to_y")
            _check(fn(self._plan, self._ptr(model, self.model_len), xs, ys, self._stream()),
                   f"sip_kkt_add_{name}_to_y")
        return y

    # ---- theta_dim > 0: r1 is [batch, x_dim + p], b / sol / x / y are [batch, full_dim] ----
    def theta_offset(self, block, index):
        off = self._lib.sip_kkt_theta_offset(self._plan, block, index)
        if off == ctypes.c_size_t(-1).value:
            raise IndexError((block, index))
        return off

    def factor_theta(self, model, theta_model, w, r1, r2, r3):
        _check(self._lib.sip_kkt_factor_theta(self._plan, self._ptr(model, self.model_len),
                                              self._ptr(theta_model, self.theta_len), self._ptr(w, self.z_dim),
                                              self._ptr(r1, self.x_dim + self.theta_dim), self._ptr(r2, self.y_dim),
                                              self._ptr(r3, self.z_dim), self.work.data_ptr(),
                                              self.theta_work.data_ptr(), self.status.data_ptr(), self._stream()),
               "sip_kkt_factor_theta")
        return self.status

    def solve_theta(self, model, theta_model, b, sol=None):
        if sol is None:
            sol = torch.zeros(self.batch, self.full_dim, dtype=torch.float64, device=self.device)
        _check(self._lib.sip_kkt_solve_theta(self._plan, self._ptr(model, self.model_len),
                                             self._ptr(theta_model, self.theta_len), self._ptr(b, self.full_dim),
                                             self._ptr(sol, self.full_dim), self.work.data_ptr(),
                                             self.theta_work.data_ptr(), self.status.data_ptr(), self._stream()),
               "sip_kkt_solve_theta")
        return sol

    def add_Kx_to_y_theta(self, model, theta_model, w, r1, r2, r3, x, y=None):
        if y is None:
            y = torch.zeros(self.batch, self.full_dim, dtype=torch.float64, device=self.device)
        _check(self._lib.sip_kkt_add_Kx_to_y_theta(self._plan, self._ptr(model, self.model_len),
                                                   self._ptr(theta_model, self.theta_len), self._ptr(w, self.z_dim),
                                                   self._ptr(r1, self.x_dim + self.theta_dim),
                                                   self._ptr(r2, self.y_dim), self._ptr(r3, self.z_dim),
                                                   self._ptr(x, self.full_dim), self._ptr(y, self.full_dim),
                                                   self._stream()), "sip_kkt_add_Kx_to_y_theta")
        return y

    def close(self):
        if self._plan:
            self._lib.sip_kkt_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
