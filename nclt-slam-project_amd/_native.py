"""ctypes binding of csrc/libreloc_hip.so (C-ABI declared in include/reloc.h).

There is deliberately no fallback: if the library is missing, cannot be loaded, or no gfx950
device is usable, every entry point raises.  Nothing here imports or calls oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RELOC_LIB: developer switch used by tools/exp_*.py to time experimental builds of the same library; like the library's own
# switches it is honoured only under RELOC_DEV=1
LIB_PATH = (os.environ.get("RELOC_LIB") if os.environ.get("RELOC_DEV") == "1" else None) or os.path.join(_HERE, "csrc", "libreloc_hip.so")

c_ctx = C.c_void_p
P = C.c_void_p
i32, i64, u64, f32, f64 = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/reloc.h one to one
SIGNATURES = {
    "reloc_last_error": (C.c_char_p, []),
    "reloc_device_count": (C.c_int, []),
    "reloc_create": (c_ctx, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "reloc_destroy": (None, [c_ctx]),
    "reloc_set_stream": (C.c_int, [c_ctx, P]),
    "reloc_sync": (C.c_int, [c_ctx]),
    "reloc_dev_alloc": (P, [c_ctx, i64]),
    "reloc_dev_free": (C.c_int, [c_ctx, P]),
    "reloc_h2d": (C.c_int, [c_ctx, P, P, i64]),
    "reloc_d2h": (C.c_int, [c_ctx, P, P, i64]),
    "reloc_timer_begin": (C.c_int, [c_ctx]),
    "reloc_timer_end": (C.c_int, [c_ctx, P]),
    "reloc_profile_enable": (C.c_int, [c_ctx, C.c_int]),
    "reloc_profile_get": (C.c_int, [c_ctx, C.c_int, P, P]),
    "reloc_gray_u8": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, C.c_int, P]),
    "reloc_orb_detect_compute": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P]),
    "reloc_orb_frame_dev": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "reloc_frame_desc_dev": (P, [c_ctx]),
    "reloc_frame_xy_dev": (P, [c_ctx]),
    "reloc_frame_count_dev": (P, [c_ctx]),
    "reloc_frame_debug_plane": (C.c_int, [c_ctx, C.c_int, C.c_int, P, P, P]),
    "reloc_record_frame": (C.c_int, [c_ctx, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P]),
    "reloc_depth_points": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, C.c_int, P, f32, f32, P, P]),
    "reloc_db_ratio_counts": (C.c_int, [c_ctx, P, C.c_int, f64, P]),
    "reloc_match_mutual": (C.c_int, [c_ctx, P, C.c_int, P, C.c_int, P, P, P, P]),
    "reloc_match_knn2": (C.c_int, [c_ctx, P, C.c_int, P, C.c_int, P, P]),
    "reloc_db_upload": (C.c_int, [c_ctx, P, P, P, P, i64]),
    "reloc_db_records": (i64, [c_ctx]),
    "reloc_db_rows": (i64, [c_ctx]),
    "reloc_db_match_counts": (C.c_int, [c_ctx, P, C.c_int, P]),
    "reloc_db_match_counts_dev": (C.c_int, [c_ctx, P, P, C.c_int, P]),
    "reloc_hamming_matrix": (C.c_int, [c_ctx, P, i64, P, i64, P]),
    "reloc_hamming_matrix_dev": (C.c_int, [c_ctx, P, i64, P, i64, P]),
    "reloc_pnp_score": (C.c_int, [c_ctx, P, P, C.c_int, P, C.c_int, P, f32, P, P]),
    "reloc_pnp_ransac": (C.c_int, [c_ctx, P, P, C.c_int, P, C.c_int, f32, f64, u64, P, P, P, P, P]),
    "reloc_set_camera": (C.c_int, [c_ctx, P, P, P]),
    "reloc_tick_debug": (C.c_int, [c_ctx, P, P, P, P, P, P, P]),
    "reloc_tick": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, P, C.c_int, u64, P, P, P, P, P, P]),
    "reloc_get_params": (C.c_int, [c_ctx, P]),
    "reloc_set_params": (C.c_int, [c_ctx, P]),
    "reloc_db_reserve": (C.c_int, [c_ctx, i64, i64]),
    "reloc_db_append": (C.c_int, [c_ctx, P, P, P, C.c_int, P, P]),
    "reloc_db_select": (C.c_int, [c_ctx, C.c_int]),
    "reloc_db_share": (C.c_int, [c_ctx, c_ctx]),
    "reloc_d2d": (C.c_int, [c_ctx, P, P, i64]),
    "reloc_host_alloc": (P, [i64]),
    "reloc_host_free": (C.c_int, [P]),
    "reloc_get_stream": (P, [c_ctx]),
    "reloc_tick_result_dev": (P, [c_ctx]),
    "reloc_tick_result_to": (C.c_int, [c_ctx, P]),
    "reloc_set_exclusive": (C.c_int, [c_ctx, C.c_int]),
    "reloc_db_fetch": (C.c_int, [c_ctx, i64, P, P, P, P, P, P]),
    "reloc_tick_result_ex": (C.c_int, [c_ctx, P, P, P, P, P, P, P, P]),
    "reloc_tick_accumulate_dev": (C.c_int, [c_ctx, P, C.c_int, C.c_int, P, C.c_int]),
    "reloc_accumulate_result": (C.c_int, [c_ctx, P, P, P]),
    "reloc_tick_dev": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, P, C.c_int, u64]),
    "reloc_tick_batch_dev": (C.c_int, [P, C.c_int, P, C.c_int, C.c_int, C.c_int, P, C.c_int, P]),
    "reloc_shard_scan_batch_dev": (C.c_int, [P, C.c_int, P, C.c_int, C.c_int, C.c_int, P, C.c_int, i64, P]),
    "reloc_shard_merge_dev": (C.c_int, [c_ctx, P, C.c_int, i64, C.c_int, C.c_int, i64, i64, P, P, P]),
    "reloc_shard_solve_batch_dev": (C.c_int, [P, C.c_int, P, C.c_int, P, P, P]),
    "reloc_tick_result": (C.c_int, [c_ctx, P, P, P, P, P, P]),
    "reloc_tick_wait": (C.c_int, [c_ctx]),
    "reloc_tick_scan_dev": (C.c_int, [c_ctx, P, C.c_int, C.c_int, C.c_int, P, P, P, C.c_int]),
    "reloc_tick_solve_dev": (C.c_int, [c_ctx, P, C.c_int, P, C.c_int, u64]),
}



class RelocParams(C.Structure):
    """mirror of `reloc_params` (include/reloc.h)"""
    _fields_ = [("nfeatures", i32), ("max_candidates", i32), ("min_matches", i32), ("min_inliers", i32),
                ("ransac_iterations", i32), ("global_max_candidates", i32), ("global_min_inliers", i32),
                ("accum_min_kpts", i32), ("candidate_radius_m", f64), ("heading_tol_deg", f64), ("reproj_max_px", f64),
                ("ransac_reproj_px", f64), ("ransac_confidence", f64), ("consistency_m", f64),
                ("global_reproj_max_px", f64), ("accum_min_dist_m", f64), ("accum_depth_min_m", f64),
                ("accum_depth_max_m", f64), ("gray_coeff_bits", i32), ("reserved0", i32)]


_lib = None


class RelocError(RuntimeError):
    """Raised for every failure of the native library (the cv2 shim re-raises it as cv2.error)."""


def _one_hip_runtime():
    """A process must run on ONE HIP runtime.  libreloc_hip.so needs `libamdhip64.so.7` and would take /opt/rocm's;
    PyTorch-ROCm ships its own copy under torch/lib with the same SONAME.  Whichever is mapped first serves both -- and
    torch on /opt/rocm's runtime reports "No HIP GPUs are available".  So, when torch is installed but not imported yet,
    its bundled runtime is mapped here first (without importing torch): afterwards the import order of torch and this
    package does not matter.  Without torch the library uses /opt/rocm's runtime.  RELOC_HIP_RUNTIME=system skips this."""
    import sys
    if os.environ.get("RELOC_HIP_RUNTIME") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):      # no torch, or a CPU-only torch: /opt/rocm's runtime is used
        pass


def load(strict: bool = True):
    """Loads the shared library and binds every symbol of include/reloc.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RelocError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  There is no CPU fallback.")
    _one_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RelocError(f"cannot load {LIB_PATH}: {e}") from e
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing and strict:
        raise RelocError(f"{LIB_PATH} lacks symbols declared in include/reloc.h: {missing}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().reloc_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RelocError(f"{what or 'libreloc_hip'} failed (code {rc}): {last_error()}")


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)
