"""Batched uniform-chain LQR on the GPU: thin torch/ctypes plumbing over the C ABI.

Mirrors the reference's `LQR` object life cycle (sip_optimal_control/lqr.hpp:
189-194): construct once per shape (topology compiled once), then
`factor_solve()` (= factor_with_status() + solve()) as often as needed on
device-resident packed buffers.  No arithmetic happens in Python.
"""
import ctypes
import enum

import torch

from ._lib import LQRLibraryError, load_library, resolve_device
from .layout import ChainShape


class FactorStatus(enum.IntEnum):
    """sip_optimal_control/lqr.hpp:68-74."""
    SUCCESS = 0
    INVALID_DELTA = 1
    F_FACTORIZATION_FAILURE = 2
    G_FACTORIZATION_FAILURE = 3
    INVALID_TOPOLOGY = 4


_DTYPES = {torch.float64: 0, torch.float32: 1}


def _check(code, what):
    if code != 0:
        names = {-1: "invalid argument", -2: "unsupported shape/dtype (no HIP kernel)",
                 -3: "HIP runtime error", -4: "allocation failure"}
        raise LQRLibraryError(f"{what} failed: {names.get(code, code)}")


class BatchedChainLQR:
    """`batch` independent chain problems of horizon T, state dim n, control dim m."""

    def __init__(self, n, m, T, batch, dtype=torch.float64, device="cuda:0", symmetric=False):
        """symmetric=True: `mats` in SIP_LQR_LAYOUT_SYMMETRIC (Q, R as packed lower triangles; ChainShape.pack_index
        converts a full-layout batch); raises for shapes without a symmetric-packed kernel."""
        self._lib = load_library()
        self.shape = ChainShape(n, m, T, symmetric=bool(symmetric))
        self.batch = int(batch)
        self.dtype = dtype
        self.device = resolve_device(device)  # explicit ordinal; raises without a HIP device
        handle = ctypes.c_void_p()
        _check(self._lib.sip_lqr_plan_create_layout(_DTYPES[dtype], self.batch, T, n, m, self.device.index,
                                                    1 if symmetric else 0, ctypes.byref(handle)),
               f"sip_lqr_plan_create_layout(n={n}, m={m}, T={T}, {dtype}, symmetric={bool(symmetric)})")
        self._plan = handle
        esize = torch.empty((), dtype=dtype).element_size()
        assert self._lib.sip_lqr_mats_len(handle) == self.shape.mats_len
        assert self._lib.sip_lqr_vecs_len(handle) == self.shape.vecs_len
        assert self._lib.sip_lqr_gains_len(handle) == self.shape.gains_len
        ws_elems = -(-self._lib.sip_lqr_workspace_bytes(handle) // esize)
        self.workspace = torch.empty(ws_elems, dtype=dtype, device=self.device)
        self.status = torch.zeros(self.batch, dtype=torch.int32, device=self.device)

    @property
    def kernel_name(self):
        return self._lib.sip_lqr_kernel_name(self._plan).decode()

    def empty_sol(self):
        return torch.empty(self.batch, self.shape.vecs_len, dtype=self.dtype, device=self.device)

    def empty_gains(self):
        return torch.empty(self.batch, self.shape.gains_len, dtype=self.dtype, device=self.device)

    def _ptr(self, t, rows, cols, name):
        if t.dtype != self.dtype or t.device != self.device or not t.is_contiguous() \
                or t.numel() != rows * cols:
            raise ValueError(f"{name}: expected contiguous {self.dtype} [{rows}, {cols}] on {self.device}")
        return ctypes.c_void_p(t.data_ptr())

    def factor_solve(self, mats, vecs, sol=None, gains=None, stream=None):
        """One fused Riccati sweep (factor + solve) over the whole batch.

        Asynchronous on `stream` (default: torch's current stream).  Returns
        (sol, gains, status) device tensors in the packed chain layout.
        """
        s = self.shape
        if sol is None:
            sol = self.empty_sol()
        if gains is None:
            gains = self.empty_gains()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        _check(self._lib.sip_lqr_factor_solve(
            self._plan, self._ptr(mats, self.batch, s.mats_len, "mats"),
            self._ptr(vecs, self.batch, s.vecs_len, "vecs"),
            self._ptr(sol, self.batch, s.vecs_len, "sol"),
            self._ptr(gains, self.batch, s.gains_len, "gains"),
            ctypes.c_void_p(self.status.data_ptr()),
            ctypes.c_void_p(self.workspace.data_ptr()),
            ctypes.c_void_p(stream.cuda_stream)), "sip_lqr_factor_solve")
        return sol, gains, self.status

    @property
    def has_split(self):
        """True when sip_lqr_factor_solve_split serves this plan (A | B read in place)."""
        return bool(self._lib.sip_lqr_has_split(self._plan))

    def split_inputs(self, mats):
        """The inputs of factor_solve_split cut out of packed `mats` [batch, mats_len]: (qmr, ab) with
        qmr [batch, split_mats_len] = node blocks [Q | delta], edge blocks [M | R], and ab [batch, T, n * (n + m)]
        = A | B per stage.  A test / demo helper: callers of the split entry point have A | B elsewhere already."""
        s = self.shape
        n, m, T = s.n, s.m, s.T
        node, edge, abn = n * n + n, n * n + 2 * n * m + m * m, n * n + n * m
        stg = node + edge
        body = mats[:, :T * stg].reshape(self.batch, T, stg)
        qmr = torch.cat([torch.cat([body[:, :, :node], body[:, :, node + abn:]], dim=2).reshape(self.batch, -1),
                         mats[:, T * stg:]], dim=1).contiguous()
        assert qmr.shape[1] == int(self._lib.sip_lqr_split_mats_len(self._plan))
        return qmr, body[:, :, node:node + abn].contiguous()

    def factor_solve_split(self, qmr, ab, vecs, sol=None, gains=None, stream=None):
        """factor_solve with the dynamics Jacobians read in place (sip_lqr_factor_solve_split): `ab` any
        tensor whose element [p, i] holds A | B of stage i (strides taken from the tensor)."""
        s = self.shape
        if sol is None:
            sol = self.empty_sol()
        if gains is None:
            gains = self.empty_gains()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        assert ab.dtype == torch.float64 and ab.stride(-1) == 1 and ab.shape[-1] >= s.n * (s.n + s.m)
        _check(self._lib.sip_lqr_factor_solve_split(
            self._plan, ctypes.c_void_p(qmr.data_ptr()), ctypes.c_void_p(ab.data_ptr()),
            ctypes.c_int64(ab.stride(0)), ctypes.c_int64(ab.stride(1) if ab.dim() > 2 else 0),
            self._ptr(vecs, self.batch, s.vecs_len, "vecs"),
            self._ptr(sol, self.batch, s.vecs_len, "sol"),
            self._ptr(gains, self.batch, s.gains_len, "gains"),
            ctypes.c_void_p(self.status.data_ptr()),
            ctypes.c_void_p(self.workspace.data_ptr()),
            ctypes.c_void_p(stream.cuda_stream)), "sip_lqr_factor_solve_split")
        return sol, gains, self.status

    def factor(self, mats, gains=None, stream=None):
        """LQR::factor_with_status() alone (general engine): statuses, K part of gains,
        factor state kept in the workspace for later solve() calls."""
        s = self.shape
        if gains is None:
            gains = self.empty_gains()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        _check(self._lib.sip_lqr_factor(
            self._plan, self._ptr(mats, self.batch, s.mats_len, "mats"),
            self._ptr(gains, self.batch, s.gains_len, "gains"),
            ctypes.c_void_p(self.status.data_ptr()), ctypes.c_void_p(self.workspace.data_ptr()),
            ctypes.c_void_p(stream.cuda_stream)), "sip_lqr_factor")
        return gains, self.status

    def solve(self, mats, vecs, gains, sol=None, stream=None):
        """LQR::solve() alone against the last factor(); repeatable with new vecs."""
        s = self.shape
        if sol is None:
            sol = self.empty_sol()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        _check(self._lib.sip_lqr_solve(
            self._plan, self._ptr(mats, self.batch, s.mats_len, "mats"),
            self._ptr(vecs, self.batch, s.vecs_len, "vecs"),
            self._ptr(sol, self.batch, s.vecs_len, "sol"),
            self._ptr(gains, self.batch, s.gains_len, "gains"),
            ctypes.c_void_p(self.workspace.data_ptr()),
            ctypes.c_void_p(stream.cuda_stream)), "sip_lqr_solve")
        return sol

    def solve_multi_workspace_bytes(self, num_rhs):
        """Column workspace sip_lqr_solve_multi needs for `num_rhs` columns; 0: the shape has no multi-rhs kernel
        and the columns are solved one by one."""
        return int(self._lib.sip_lqr_solve_multi_workspace_bytes(self._plan, int(num_rhs)))

    def solve_multi(self, mats, vecs_cols, gains, sol_cols=None, stream=None):
        """LQR::solve() for several right-hand sides against the last factor() (the multi-rhs block of
        solve_stagewise_kkt_matrix, helpers.cpp:521-665): vecs_cols / sol_cols are [num_rhs, batch,
        vecs_len].  One sweep per 8 columns where the shape has the multi-rhs kernel."""
        s = self.shape
        num_rhs = vecs_cols.shape[0]
        if sol_cols is None:
            sol_cols = torch.empty(num_rhs, self.batch, s.vecs_len, dtype=self.dtype, device=self.device)
        for t, name in ((vecs_cols, "vecs_cols"), (sol_cols, "sol_cols")):
            if t.dtype != self.dtype or t.device != self.device or not t.is_contiguous() or \
                    tuple(t.shape) != (num_rhs, self.batch, s.vecs_len):
                raise ValueError(f"{name}: expected contiguous {self.dtype} [{num_rhs}, {self.batch}, {s.vecs_len}]")
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        need = self._lib.sip_lqr_solve_multi_workspace_bytes(self._plan, num_rhs)
        if getattr(self, "_col_ws", None) is None or self._col_ws.numel() < need:
            self._col_ws = torch.empty(max(1, need), dtype=torch.uint8, device=self.device)
        _check(self._lib.sip_lqr_solve_multi(
            self._plan, self._ptr(mats, self.batch, s.mats_len, "mats"), ctypes.c_void_p(vecs_cols.data_ptr()),
            ctypes.c_void_p(sol_cols.data_ptr()), num_rhs, self._ptr(gains, self.batch, s.gains_len, "gains"),
            ctypes.c_void_p(self.workspace.data_ptr()), ctypes.c_void_p(self._col_ws.data_ptr()),
            ctypes.c_void_p(stream.cuda_stream)), "sip_lqr_solve_multi")
        return sol_cols

    def close(self):
        if getattr(self, "_plan", None):
            self._lib.sip_lqr_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
