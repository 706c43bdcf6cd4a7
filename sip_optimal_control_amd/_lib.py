"""ctypes binding of the C ABI declared in include/sip_lqr_amd.h and
include/sip_kkt_amd.h.

Fails loudly (LQRLibraryError) when the HIP library is missing: the product
has no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsip_lqr_amd.so"


class LQRLibraryError(RuntimeError):
    pass


def library_path():
    # SIP_LQR_LIB: diagnostic builds (tools/diag_build.sh) only.
    return os.environ.get("SIP_LQR_LIB") or os.path.join(_HERE, "lib", _LIB_NAME)


# name -> (restype, argtypes); mirrors include/sip_lqr_amd.h one to one.
_P = ctypes.c_void_p
_PP = ctypes.POINTER(ctypes.c_void_p)
_SIGNATURES = {
    "sip_lqr_plan_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _PP]),
    "sip_lqr_plan_create_layout": (ctypes.c_int, [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, _PP]),
    "sip_lqr_plan_layout": (ctypes.c_int, [_P]),
    "sip_lqr_plan_destroy": (None, [_P]),
    "sip_lqr_mats_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_vecs_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_sol_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_gains_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_status_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_workspace_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_plan_batch": (ctypes.c_int64, [_P]),
    "sip_lqr_scalar_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_mats_len": (ctypes.c_size_t, [_P]),
    "sip_lqr_vecs_len": (ctypes.c_size_t, [_P]),
    "sip_lqr_gains_len": (ctypes.c_size_t, [_P]),
    "sip_lqr_pack_problem": (ctypes.c_int, [_P, ctypes.c_int64] + [_P] * 9 + [_P, _P]),
    "sip_lqr_unpack_solution": (ctypes.c_int, [_P, ctypes.c_int64, _P, _P, _P, _P]),
    "sip_lqr_unpack_gains": (ctypes.c_int, [_P, ctypes.c_int64, _P, _P, _P]),
    "sip_lqr_factor_solve": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sip_lqr_has_split": (ctypes.c_int, [_P]),
    "sip_lqr_split_mats_len": (ctypes.c_int64, [_P]),
    "sip_lqr_factor_solve_split": (ctypes.c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int64, _P, _P, _P, _P, _P, _P]),
    "sip_lqr_factor": (ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    "sip_lqr_solve": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "sip_lqr_solve_multi_workspace_bytes": (ctypes.c_size_t, [_P, ctypes.c_int]),
    "sip_lqr_solve_multi": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, _P, _P, _P, _P]),
    "sip_lqr_compile_topology": (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [_P] * 9),
    "sip_lqr_tree_plan_create": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P,
                                                ctypes.c_int, _PP]),
    "sip_lqr_tree_plan_destroy": (None, [_P]),
    "sip_lqr_tree_topology_status": (ctypes.c_int, [_P]),
    "sip_lqr_tree_topology_array": (ctypes.POINTER(ctypes.c_int), [_P, ctypes.c_int]),
    "sip_lqr_tree_input_len": (ctypes.c_size_t, [_P]),
    "sip_lqr_tree_work_len": (ctypes.c_size_t, [_P]),
    "sip_lqr_tree_output_len": (ctypes.c_size_t, [_P]),
    "sip_lqr_tree_offset": (ctypes.c_size_t, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sip_lqr_tree_factor": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "sip_lqr_tree_solve": (ctypes.c_int, [_P, _P, _P, _P, _P, _P]),
    "sip_lqr_tree_fused_scratch_bytes": (ctypes.c_size_t, [_P]),
    "sip_lqr_tree_factor_solve": (ctypes.c_int, [_P] * 7),
    "sip_lqr_tree_factor_solve_workspace": (ctypes.c_int, [_P] * 7),
    "sip_lqr_tree_kernel_name": (ctypes.c_char_p, [_P]),
    "sip_lqr_kernel_name": (ctypes.c_char_p, [_P]),
    "sip_lqr_version": (ctypes.c_char_p, []),
    # include/sip_kkt_amd.h
    "sip_kkt_plan_create": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [_P] * 8 +
                            [ctypes.c_int, _PP]),
    "sip_kkt_plan_destroy": (None, [_P]),
    "sip_kkt_input_status": (ctypes.c_int, [_P]),
    "sip_kkt_len": (ctypes.c_size_t, [_P, ctypes.c_int]),
    "sip_kkt_model_offset": (ctypes.c_size_t, [_P, ctypes.c_int, ctypes.c_int]),
    "sip_kkt_vector_offset": (ctypes.c_size_t, [_P, ctypes.c_int, ctypes.c_int]),
    "sip_kkt_work_bytes": (ctypes.c_size_t, [_P]),
    "sip_kkt_kernel_name": (ctypes.c_char_p, [_P]),
    "sip_kkt_factor": (ctypes.c_int, [_P] * 9),
    "sip_kkt_solve": (ctypes.c_int, [_P] * 7),
    "sip_kkt_factor_solve": (ctypes.c_int, [_P] * 11),
    "sip_kkt_add_Kx_to_y": (ctypes.c_int, [_P] * 9),
    "sip_kkt_plan_set_theta": (ctypes.c_int, [_P, ctypes.c_int]),
    "sip_kkt_theta_len": (ctypes.c_size_t, [_P]),
    "sip_kkt_theta_offset": (ctypes.c_size_t, [_P, ctypes.c_int, ctypes.c_int]),
    "sip_kkt_theta_work_bytes": (ctypes.c_size_t, [_P]),
    "sip_kkt_factor_theta": (ctypes.c_int, [_P] * 11),
    "sip_kkt_solve_theta": (ctypes.c_int, [_P] * 9),
    "sip_kkt_add_Kx_to_y_theta": (ctypes.c_int, [_P] * 10),
}
for _op in ("Hx", "Cx", "CTx", "Gx", "GTx"):  # the five block operators, helpers.hpp:20-24
    _SIGNATURES[f"sip_kkt_add_{_op}_to_y"] = (ctypes.c_int, [_P] * 5)
    _SIGNATURES[f"sip_kkt_add_{_op}_to_y_theta"] = (ctypes.c_int, [_P] * 6)

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def resolve_device(device):
    """torch.device with an explicit ordinal: a bare "cuda" means the CURRENT device (as everywhere in
    torch), not device 0.  The C ABI launches on the plan's ordinal whatever the caller's current
    device is (DeviceGuard, csrc/stream_fill.hpp), so buffers, streams and kernels agree."""
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        raise LQRLibraryError(f"{device!r}: the batched solvers need a HIP device; there is no CPU path")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev

_lib = None


def load_library():
    """Load lib/libsip_lqr_amd.so and bind every symbol of the header."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise LQRLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise LQRLibraryError(f"cannot load {path}: {exc}") from exc
    for name, (restype, argtypes) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise LQRLibraryError(f"{path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
