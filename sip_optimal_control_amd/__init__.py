"""MI355X-native batched regularized-LQR / Riccati solve.

Drop-in for the Newton-KKT linear solve of joaospinto/sip_optimal_control
(`sip_optimal_control/lqr.{hpp,cpp}`): hand-written gfx950 HIP kernels behind a
C ABI (`include/sip_lqr_amd.h`, built into `lib/libsip_lqr_amd.so`).  This
Python package is plumbing only (device buffers via torch, ctypes calls into
the C ABI); there is no CPU compute path -- importing works without a GPU,
every solve requires the HIP library and a device.
"""
from ._lib import LQRLibraryError, load_library, library_path  # noqa: F401
from .layout import ChainShape  # noqa: F401
from .chain import BatchedChainLQR, FactorStatus  # noqa: F401
from .kkt import BatchedNewtonKKT  # noqa: F401
from . import synthetic  # noqa: F401

__all__ = [
    "BatchedChainLQR",
    "BatchedNewtonKKT",
    "ChainShape",
    "FactorStatus",
    "LQRLibraryError",
    "load_library",
    "library_path",
    "synthetic",
]
