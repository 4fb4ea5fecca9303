"""ctypes binding of include/gpcc_attr_mi355.h.  There is NO CPU fallback:
if the HIP library is missing or a call fails this module raises."""
import ctypes as C
import os

from .params import LiftParams, LodParams, PredParams, RahtParams

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# GPCC_LIB_PATH: an experiment build of the same library (tools/, parameter sweeps)
LIB_PATH = os.environ.get("GPCC_LIB_PATH") or os.path.join(PKG_DIR, "libgpcc_attr_mi355.so")

ABI_VERSION = 6  # GPCC_ABI_VERSION of include/gpcc_attr_mi355.h

# every symbol the header declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "gpcc_raht_set_prediction_weights", "gpcc_abi_version", "gpcc_last_error", "gpcc_clear_last_error",
    "gpcc_device_count", "gpcc_ctx_create", "gpcc_ctx_destroy",
    "gpcc_ctx_synchronize", "gpcc_ctx_workspace_bytes", "gpcc_ctx_set_morton_bits", "gpcc_ctx_set_fast_arith", "gpcc_ctx_pred_pass_stats",
    "gpcc_raht_forward", "gpcc_raht_inverse", "gpcc_attr_morton_sort",
    "gpcc_raht_forward_inter", "gpcc_raht_inverse_inter",
    "gpcc_dev_raht_forward", "gpcc_dev_raht_inverse", "gpcc_dev_attr_morton_sort",
    "gpcc_ctx_set_profiling", "gpcc_ctx_kernel_times", "gpcc_ctx_stats",
    "gpcc_lift_forward", "gpcc_lift_inverse", "gpcc_lod_compute_weights", "gpcc_lod_build", "gpcc_lod_build_inter", "gpcc_lift_forward_inter", "gpcc_lift_inverse_inter", "gpcc_pred_forward_inter", "gpcc_pred_inverse_inter", "gpcc_estimate_dist2", "gpcc_recolour", "gpcc_raht_encode_attr", "gpcc_raht_decode_attr",
    "gpcc_lift_encode_attr", "gpcc_lift_decode_attr", "gpcc_zero_run_pack", "gpcc_raht_encode_attr_packed",
    "gpcc_raht_encode_attr_packed_regions", "gpcc_raht_decode_attr_regions",
    "gpcc_dev_lod_build", "gpcc_dev_lift_encode_attr", "gpcc_dev_lift_decode_attr",
    "gpcc_multi_create", "gpcc_multi_destroy", "gpcc_multi_num_devices", "gpcc_multi_uses_rccl", "gpcc_multi_rccl_selftest",
    "gpcc_multi_raht_forward", "gpcc_multi_raht_inverse", "gpcc_binarise_symbols",
    "gpcc_multi_lift_encode_attr", "gpcc_multi_lift_decode_attr", "gpcc_multi_pred_encode_attr", "gpcc_multi_pred_decode_attr",
    "gpcc_pred_forward", "gpcc_pred_inverse", "gpcc_pred_encode_attr", "gpcc_pred_decode_attr",
    "gpcc_dev_pred_encode_attr", "gpcc_dev_pred_decode_attr",
    "gpcc_ctx_reserve",
    "gpcc_debug_alloc_events", "gpcc_debug_has_experiments", "gpcc_debug_guard_checks", "gpcc_debug_rate_sum",
    "gpcc_debug_guard_selftest",
]


class GpccError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gpcc error {code}: {msg}")
        self.code = code


class CtxStats(C.Structure):
    _fields_ = [("calls_ok", C.c_int64), ("calls_unsupported", C.c_int64),
                ("calls_failed", C.c_int64), ("points_ok", C.c_int64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("total_ms", C.c_double), ("launches", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m __graft_entry__` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64
    # (SONAME libamdhip64.so.7, same as /opt/rocm's).  Importing torch first
    # makes our NEEDED entry bind to the copy torch already mapped; the
    # other order maps two runtimes and torch then sees "No HIP GPUs".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    pp = C.POINTER(RahtParams)
    lib.gpcc_abi_version.restype = C.c_int
    # the ctypes parameter blocks of params.py mirror THIS version of the header
    if lib.gpcc_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI {lib.gpcc_abi_version()}, the Python mirror expects {ABI_VERSION}: rebuild")
    lib.gpcc_last_error.restype = C.c_char_p
    lib.gpcc_device_count.restype = C.c_int
    lib.gpcc_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.gpcc_ctx_destroy.argtypes = [vp]
    lib.gpcc_ctx_destroy.restype = None
    lib.gpcc_ctx_synchronize.argtypes = [vp]
    lib.gpcc_ctx_workspace_bytes.argtypes = [vp]
    lib.gpcc_ctx_workspace_bytes.restype = C.c_size_t
    lib.gpcc_ctx_set_morton_bits.argtypes = [vp, i32]
    lib.gpcc_ctx_reserve.argtypes = [vp, C.c_int64, i32, i32]
    lib.gpcc_ctx_set_fast_arith.argtypes = [vp, i32]
    lib.gpcc_ctx_pred_pass_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.gpcc_ctx_set_profiling.argtypes = [vp, C.c_int]
    lib.gpcc_ctx_kernel_times.argtypes = [vp, C.POINTER(KernelTime), i32]
    lib.gpcc_ctx_stats.argtypes = [vp, C.POINTER(CtxStats)]
    lib.gpcc_raht_set_prediction_weights.argtypes = [pp, C.POINTER(i32)]
    lib.gpcc_raht_set_prediction_weights.restype = None
    for name in ("gpcc_raht_forward", "gpcc_raht_inverse"):
        getattr(lib, name).argtypes = [vp, pp, vp, vp, vp, vp, i32, i32]
    lib.gpcc_raht_forward_inter.argtypes = [vp, pp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, C.POINTER(i32), vp, C.POINTER(i32)]
    lib.gpcc_raht_inverse_inter.argtypes = [vp, pp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, i32, vp, i32]
    lib.gpcc_attr_morton_sort.argtypes = [vp, vp, i32, vp, vp]
    for name in ("gpcc_dev_raht_forward", "gpcc_dev_raht_inverse"):
        getattr(lib, name).argtypes = [vp, pp, i32, i64p, vp, vp, vp, vp, i32]
    lib.gpcc_dev_attr_morton_sort.argtypes = [vp, i32, i64p, vp, vp, vp]
    for name in ("gpcc_lift_forward", "gpcc_lift_inverse"):
        getattr(lib, name).argtypes = [vp, C.POINTER(LiftParams), i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    for name in ("gpcc_pred_forward", "gpcc_pred_inverse"):
        getattr(lib, name).argtypes = [vp, C.POINTER(PredParams), i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    for name in ("gpcc_pred_encode_attr", "gpcc_pred_decode_attr"):
        getattr(lib, name).argtypes = [vp, C.POINTER(LodParams), C.POINTER(PredParams), vp, vp, vp, vp, vp, i32, i32]
    for name in ("gpcc_dev_pred_encode_attr", "gpcc_dev_pred_decode_attr"):
        getattr(lib, name).argtypes = [vp, C.POINTER(LodParams), vp, i32, i64p, vp, vp, vp, vp, vp, i32]
    lib.gpcc_lod_compute_weights.argtypes = [vp, i32, vp, vp, vp]
    lib.gpcc_lod_build.argtypes = [vp, C.POINTER(LodParams), vp, i32, vp, vp, vp, vp, vp, C.POINTER(i32)]
    for f in (lib.gpcc_lift_forward_inter, lib.gpcc_lift_inverse_inter):
        f.argtypes = [vp, C.POINTER(LiftParams), i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    for f in (lib.gpcc_pred_forward_inter, lib.gpcc_pred_inverse_inter):
        f.argtypes = [vp, C.POINTER(PredParams), i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.gpcc_lod_build_inter.argtypes = [vp, C.POINTER(LodParams), vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp,
                                         C.POINTER(i32), vp]
    for name in ("gpcc_raht_encode_attr", "gpcc_raht_decode_attr"):
        getattr(lib, name).argtypes = [vp, pp, vp, vp, vp, i32, i32, i32]
    for name in ("gpcc_lift_encode_attr", "gpcc_lift_decode_attr"):
        getattr(lib, name).argtypes = [vp, C.POINTER(LodParams), C.POINTER(LiftParams), vp, vp, vp, vp, vp, i32, i32]
    lib.gpcc_raht_encode_attr_packed.argtypes = [vp, pp, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32), i32, i32, i32]
    lib.gpcc_raht_encode_attr_packed_regions.argtypes = [vp, pp, vp, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32), i32, i32, i32]
    lib.gpcc_raht_decode_attr_regions.argtypes = [vp, pp, vp, vp, vp, vp, i32, i32, i32]
    lib.gpcc_zero_run_pack.argtypes = [vp, vp, i32, i32, i32, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    lib.gpcc_estimate_dist2.argtypes = [vp, vp, i32, i32, i32, C.c_float, C.POINTER(i32)]
    lib.gpcc_recolour.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, C.c_float, C.POINTER(i32), vp]
    lib.gpcc_dev_lod_build.argtypes = [vp, C.POINTER(LodParams), i32, i64p, vp, vp, vp, vp, vp, vp, vp]
    for name in ("gpcc_dev_lift_encode_attr", "gpcc_dev_lift_decode_attr"):
        getattr(lib, name).argtypes = [vp, C.POINTER(LodParams), vp, i32, i64p, vp, vp, vp, vp, vp, i32]
    lib.gpcc_multi_create.argtypes = [C.POINTER(i32), i32, C.POINTER(vp)]
    lib.gpcc_multi_destroy.argtypes = [vp]
    lib.gpcc_multi_destroy.restype = None
    lib.gpcc_multi_num_devices.argtypes = [vp]
    lib.gpcc_multi_uses_rccl.argtypes = [vp]
    for name in ("gpcc_multi_raht_forward", "gpcc_multi_raht_inverse"):
        getattr(lib, name).argtypes = [vp, pp, i32, i64p, vp, vp, vp, i32]
    for name in ("gpcc_multi_lift_encode_attr", "gpcc_multi_lift_decode_attr", "gpcc_multi_pred_encode_attr",
                 "gpcc_multi_pred_decode_attr"):
        getattr(lib, name).argtypes = [vp, C.POINTER(LodParams), vp, i32, i64p, vp, vp, vp, vp, vp, i32]
    lib.gpcc_binarise_symbols.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.c_int64, C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise GpccError(code, load().gpcc_last_error().decode())
