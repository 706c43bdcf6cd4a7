"""Runs the C++ re-expression of the reference's tests/lqr_test.cpp against the
drop-in `sip::optimal_control::LQR` adapter (include/sip_optimal_control_amd/
lqr_dropin.hpp) on the GPU: tests/cpp/test_dropin.cpp."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("engine", ["fused (default)", "general"])
def test_cpp_dropin_suite(engine):
    """The reference's 14 tests + the adapter's own: once as a user gets the class (trees and chains on the fused
    size-class kernel wherever it applies, every LQR::Workspace field written out) and once held to the general
    engine (SIP_LQR_DROPIN_GENERAL=1)."""
    import __graft_entry__ as entry
    entry.build_hip()
    exe = entry.build_dropin_test()
    env = dict(os.environ)
    env.pop("SIP_LQR_DROPIN_GENERAL", None)
    if engine == "general":
        env["SIP_LQR_DROPIN_GENERAL"] = "1"
    proc = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "0 failures" in proc.stdout
    assert proc.stdout.count("[  OK  ]") >= 19


def test_cpp_callback_provider_suite():
    """tests/variable_dimensions_test.cpp's CallbackProvider tests (chain, sibling edges,
    zero-dimensional root, Schur variables) against the drop-in CallbackProvider class."""
    import __graft_entry__ as entry
    entry.build_hip()
    exe = entry.build_callback_provider_test()
    proc = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-3000:]
    assert "0 failures" in proc.stdout
    assert proc.stdout.count("[  OK  ]") == 5


def test_cpp_group_all_gather():
    """include/sip_lqr_amd_rccl.h: communicators over the devices present (one here), a sweep per
    device, all-gather of the gains compared bitwise (tests/cpp/test_group.cpp)."""
    import __graft_entry__ as entry
    entry.build_hip()
    entry.build_rccl()
    exe = entry.build_group_test()
    proc = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "0 failures" in proc.stdout
