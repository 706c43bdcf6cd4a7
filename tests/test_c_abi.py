"""The C-ABI shared library loads and exports every symbol include/*.h declares;
host-only entry points (plan, sizes, pack/unpack) behave; no compute calls here."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build_hip()
    from sip_optimal_control_amd._lib import load_library
    return load_library()


def _declared_functions(rccl=False):
    """Entry points declared by the headers of libsip_lqr_amd.so (rccl=False) or of the separate
    communication library libsip_lqr_amd_rccl.so (rccl=True)."""
    names = set()
    for header in glob.glob(os.path.join(ROOT, "include", "*.h")):
        if header.endswith("_rccl.h") != rccl:
            continue
        text = open(header).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(sip_(?:lqr|kkt)_\w+)\s*\(", text))
    return names


def test_exports_every_declared_symbol(lib):
    from sip_optimal_control_amd import _lib
    declared = _declared_functions()
    assert declared, "no declarations found"
    raw = ctypes.CDLL(_lib.library_path())
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)


def test_plan_sizes_and_errors(lib):
    from sip_optimal_control_amd import ChainShape
    h = ctypes.c_void_p()
    assert lib.sip_lqr_plan_create(0, 4096, 50, 12, 4, 0, ctypes.byref(h)) == 0
    s = ChainShape(12, 4, 50)
    assert lib.sip_lqr_mats_len(h) == s.mats_len == 20756
    assert lib.sip_lqr_vecs_len(h) == s.vecs_len == 1424
    assert lib.sip_lqr_gains_len(h) == s.gains_len == 2600
    assert lib.sip_lqr_mats_bytes(h) == 4096 * 20756 * 8
    assert lib.sip_lqr_sol_bytes(h) == 4096 * 1424 * 8
    assert lib.sip_lqr_status_bytes(h) == 4096 * 4
    assert lib.sip_lqr_workspace_bytes(h) > 0
    # 209 632 algorithmic bytes per sweep at this shape
    assert s.algorithmic_bytes(8) == 209632
    assert b"qw16" in lib.sip_lqr_kernel_name(h)
    lib.sip_lqr_plan_destroy(h)
    # shapes without a dedicated kernel go to the general GPU engine (never to the host)
    assert lib.sip_lqr_plan_create(0, 8, 5, 35, 3, 0, ctypes.byref(h)) == 0
    assert b"tree_generic" in lib.sip_lqr_kernel_name(h)
    lib.sip_lqr_plan_destroy(h)
    assert lib.sip_lqr_plan_create(1, 8, 100, 32, 8, 0, ctypes.byref(h)) == 0   # C4 shape, fp32
    assert lib.sip_lqr_mats_bytes(h) + lib.sip_lqr_vecs_bytes(h) == 8 * 273920 * 4
    lib.sip_lqr_plan_destroy(h)
    os.environ["SIP_LQR_VARIANT"] = "no-such-variant"
    try:
        assert lib.sip_lqr_plan_create(0, 8, 5, 12, 4, 0, ctypes.byref(h)) == -2
    finally:
        del os.environ["SIP_LQR_VARIANT"]
    assert lib.sip_lqr_plan_create(0, 0, 5, 12, 4, 0, ctypes.byref(h)) == -1
    assert lib.sip_lqr_plan_create(7, 8, 5, 12, 4, 0, ctypes.byref(h)) == -1


def test_kernel_selection_and_embedding(lib, monkeypatch):
    """Host-side dispatch (no GPU needed): exact kernels for the reference's benchmark grid, the
    smallest larger fused kernel for other uniform shapes, the general engine beyond."""
    h = ctypes.c_void_p()

    def name(n, m, dtype=0):
        assert lib.sip_lqr_plan_create(dtype, 16, 10, n, m, 0, ctypes.byref(h)) == 0
        out = lib.sip_lqr_kernel_name(h).decode()
        ws = lib.sip_lqr_workspace_bytes(h)
        lib.sip_lqr_plan_destroy(h)
        return out, ws

    for n in (4, 6, 8, 12):
        for m in (1, 2, 3, 4):
            kernel, _ = name(n, m)
            assert f"qw16<{n},{m}," in kernel and "embedding" not in kernel
            assert "staged" in kernel  # odd m too: pieces from 8-byte-aligned sources, gains by dwords
    # every fp64 shape n <= 16, m <= 8 has an exact kernel (qw16_extra.hip); LDS-staged wherever its images fit a
    # workgroup's 64 KiB: all n <= 15; n = 16 (distributed-vector mode) stays on direct loads
    for n in range(1, 17):
        for m in range(1, 9):
            kernel, _ = name(n, m)
            assert f"qw16<{n},{m}," in kernel and "embedding" not in kernel, (n, m, kernel)
            assert ("staged" in kernel) == (n <= 15), (n, m, kernel)
    # 16 < n <= 32, m <= 8: the n = 32 matrix-core kernel (chain_mt16.hpp), exact at (32, 4) and (32, 8), an
    # embedding below; beyond that the general engine
    assert name(32, 8)[0] == "chain_factor_solve_mt16<32,8,mfma16x16x4>/f64"
    assert name(32, 4)[0] == "chain_factor_solve_mt16<32,4,mfma16x16x4>/f64"
    assert name(17, 4)[0] == "chain_factor_solve_mt16<32,4,mfma16x16x4>/f64 embedding (17,4)"
    assert name(31, 5)[0] == "chain_factor_solve_mt16<32,8,mfma16x16x4>/f64 embedding (31,5)"
    assert "tree_generic" in name(33, 4)[0] and "tree_generic" in name(12, 9)[0]
    assert name(32, 8, dtype=1)[0] == "chain_factor_solve_mt16<32,8,mfma16x16x4>/f32"
    # without the extra slices (diagnostic builds; SIP_LQR_EXTRA=0): embedding in the next larger kernel
    monkeypatch.setenv("SIP_LQR_EXTRA", "0")
    assert name(10, 3)[0] == "chain_factor_solve_qw16<12,3,staged>/f64 embedding (10,3)"
    assert name(5, 3)[0].startswith("chain_factor_solve_qw16<6,3,staged>")
    assert name(13, 5)[0].startswith("chain_factor_solve_qw16<14,8,staged>")
    assert name(15, 7)[0].startswith("chain_factor_solve_qw16<15,8,staged>")
    assert name(15, 3)[0].startswith("chain_factor_solve_qw16<15,4,staged>")
    assert "qw16<16,4,direct>" in name(16, 4)[0]  # distributed-vector mode
    assert "mt16" in name(17, 4)[0] and "tree_generic" in name(12, 9)[0]
    assert "tree_generic" in name(10, 3, dtype=1)[0]  # fp32: only the n = 32 kernel is dedicated
    exact_ws, embedded_ws = name(12, 3)[1], name(10, 3)[1]
    assert embedded_ws > exact_ws  # padded copies of mats / vecs / sol / gains live in the workspace
    monkeypatch.setenv("SIP_LQR_PAD", "0")
    assert "tree_generic" in name(10, 3)[0]


def _tables(blocks):
    keep, tabs = [], {}
    for name, arrs in blocks.items():
        flat = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F")) for a in arrs]
        tab = (ctypes.c_void_p * max(1, len(flat)))(*[f.ctypes.data for f in flat])
        keep.append(flat)
        tabs[name] = tab
    return tabs, keep


def test_pack_unpack_roundtrip(lib):
    """sip_lqr_pack_problem lays the reference's double** blocks (lqr.hpp:76-85)
    out exactly as the documented packed chain layout."""
    from oracle import dense_kkt
    from sip_optimal_control_amd import ChainShape, synthetic
    n, m, T, batch = 4, 2, 5, 3
    shape = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(shape, batch, seed=4, cross_term=0.1)
    h = ctypes.c_void_p()
    assert lib.sip_lqr_plan_create(0, batch, T, n, m, 0, ctypes.byref(h)) == 0
    out_m = np.zeros((batch, shape.mats_len))
    out_v = np.zeros((batch, shape.vecs_len))
    for p in range(batch):
        blocks = dense_kkt.chain_blocks_from_packed(n, m, T, mats[p].numpy(), vecs[p].numpy())
        tabs, keep = _tables(blocks)
        rc = lib.sip_lqr_pack_problem(h, p, tabs["Q"], tabs["M"], tabs["R"], tabs["q"], tabs["r"],
                                      tabs["A"], tabs["B"], tabs["c"], tabs["delta"],
                                      out_m.ctypes.data, out_v.ctypes.data)
        assert rc == 0
    np.testing.assert_array_equal(out_m, mats.numpy())
    np.testing.assert_array_equal(out_v, vecs.numpy())
    # unpack: sol buffer -> x/u/y tables, gains -> K/k tables
    sol = np.arange(batch * shape.vecs_len, dtype=np.float64).reshape(batch, -1)
    x = [np.zeros(n) for _ in range(T + 1)]
    y = [np.zeros(n) for _ in range(T + 1)]
    u = [np.zeros(m) for _ in range(T)]
    tx = (ctypes.c_void_p * (T + 1))(*[a.ctypes.data for a in x])
    ty = (ctypes.c_void_p * (T + 1))(*[a.ctypes.data for a in y])
    tu = (ctypes.c_void_p * T)(*[a.ctypes.data for a in u])
    assert lib.sip_lqr_unpack_solution(h, 1, sol.ctypes.data, tx, tu, ty) == 0
    xs, us, ys = dense_kkt.chain_sol_from_packed(n, m, T, sol[1])
    for a, b in list(zip(x, xs)) + list(zip(u, us)) + list(zip(y, ys)):
        np.testing.assert_array_equal(a, b)
    gains = np.arange(batch * shape.gains_len, dtype=np.float64).reshape(batch, -1)
    K = [np.zeros(m * n) for _ in range(T)]
    k = [np.zeros(m) for _ in range(T)]
    tK = (ctypes.c_void_p * T)(*[a.ctypes.data for a in K])
    tk = (ctypes.c_void_p * T)(*[a.ctypes.data for a in k])
    assert lib.sip_lqr_unpack_gains(h, 2, gains.ctypes.data, tK, tk) == 0
    off = 0
    for e in range(T):
        np.testing.assert_array_equal(K[e], gains[2, off:off + m * n]); off += m * n
        np.testing.assert_array_equal(k[e], gains[2, off:off + m]); off += m
    # argument checking
    assert lib.sip_lqr_unpack_gains(h, 99, gains.ctypes.data, tK, tk) == -1
    lib.sip_lqr_plan_destroy(h)


def test_no_cpu_compute_path():
    """The product refuses to run without a HIP device instead of falling back."""
    import torch
    from sip_optimal_control_amd import BatchedChainLQR, LQRLibraryError
    with pytest.raises(LQRLibraryError):
        BatchedChainLQR(12, 4, 50, 8, device="cpu")
    if not torch.cuda.is_available():
        import sip_optimal_control_amd, inspect
        src = inspect.getsource(sip_optimal_control_amd.chain)
        assert "oracle" not in src


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sip_optimal_control_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "lqr_oracle" not in text and "from oracle" not in text and \
                    "import oracle" not in text, f


def test_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: both headers must compile as C99 (no C++-only constructs)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "sip_lqr_amd.h"\n#include "sip_kkt_amd.h"\n'
                   "int main(void) { return (int)sizeof(sip_lqr_plan *) - (int)sizeof(sip_kkt_plan *); }\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


def test_rccl_library_exports_its_header(lib):
    """include/sip_lqr_amd_rccl.h lives in its own library (links librccl; never loaded by the
    Python package, which exchanges through torch.distributed)."""
    import __graft_entry__ as entry
    path = entry.build_rccl()
    declared = _declared_functions(rccl=True)
    assert declared == {"sip_lqr_group_create", "sip_lqr_group_destroy", "sip_lqr_group_size",
                        "sip_lqr_group_all_gather_gains", "sip_lqr_all_gather_gains",
                        "sip_lqr_gains_chunk_range", "sip_lqr_gains_chunk_offset", "sip_lqr_all_gather_gains_chunk",
                        "sip_lqr_group_all_gather_gains_chunk", "sip_lqr_group_all_gather_gains_p2p",
                        "sip_lqr_group_all_gather_gains_p2p_chunk"}
    out = __import__("subprocess").run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    for name in declared:
        assert f" T {name}" in out, name
    package = os.path.join(ROOT, "sip_optimal_control_amd")
    for f in glob.glob(os.path.join(package, "*.py")):
        assert "amd_rccl" not in open(f).read()
