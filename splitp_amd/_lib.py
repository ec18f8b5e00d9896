"""ctypes binding of libsplitp_hip.so (C ABI: include/splitp_hip.h).

There is NO CPU fallback: if the library is missing, or no HIP device is visible, every
entry point of the package that needs a kernel raises `SplitPDeviceError`."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPLITP_LIB: load another build of the same library instead, e.g. `make -C splitp_amd/csrc asan` - host code under
# AddressSanitizer, device code unchanged)
LIB_PATH = os.environ.get("SPLITP_LIB") or os.path.join(_HERE, "libsplitp_hip.so")

SP_METHOD_FLATTENING = 0
SP_METHOD_SUBFLATTENING = 1
SP_METHOD_FLATTENING_DENSE = 2
SP_METHOD_FLATTENING_SPARSE = 3
SP_METHOD_MUTUAL_INFORMATION = 4
SP_N_PHASES = 11
PHASE_NAMES = ("reindex", "scatter", "gram", "eigen", "moment", "subscore", "hist", "dense", "sparse", "divergence", "chain")
SP_OK, SP_EINVAL, SP_EHIP, SP_ENOMEM, SP_ELIMIT, SP_ENOCONV = 0, 1, 2, 3, 4, 5
SP_ABI_VERSION = 4

# every symbol include/splitp_hip.h declares
SYMBOLS = (
    "sp_abi_version", "sp_last_error", "sp_device_count",
    "sp_ctx_create", "sp_ctx_destroy", "sp_ctx_set_stream", "sp_ctx_synchronize", "sp_ctx_set_gram_mode",
    "sp_ctx_set_option", "sp_ctx_get_option",
    "sp_ctx_enable_timing", "sp_ctx_reset_timing", "sp_ctx_phase_times",
    "sp_alignment_create", "sp_alignment_from_sequences", "sp_alignment_from_site_keys", "sp_simulate_alignment",
    "sp_alignment_destroy", "sp_alignment_info", "sp_alignment_fetch",
    "sp_flatten_indices", "sp_flatten_reduced_prepare", "sp_flatten_reduced_fetch", "sp_flatten_dense_counts",
    "sp_subflatten", "sp_moment_matrix",
    "sp_score_matrix_f64", "sp_score_coo_f64", "sp_divergence_matrix_f64", "sp_score_splits", "sp_score_splits_async",
    "sp_score_splits_multi_async", "sp_score_all_splits", "sp_score_all_splits_shard",
    "sp_plan_create", "sp_plan_retain", "sp_plan_release", "sp_plan_info", "sp_score_plan_async", "sp_score_plan_steps",
    "sp_finish_flagged", "sp_debug_radix_sort",
    "sp_node_create", "sp_node_destroy", "sp_node_info", "sp_node_score_all_splits",
)


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the shared library is built from (csrc/*.hip, csrc/*.h and the
    ABI header, in name order).  tools/pmc_summarise.py stamps every PMC record under profiles/ with it and bench.py
    prints a counter-based roofline fraction only when the stamp equals the hash of the sources in the tree - a counter
    profile of an older kernel must not describe a newer one."""
    import glob
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "splitp_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class SplitPDeviceError(RuntimeError):
    """The HIP library is missing / failed, or no GPU is visible."""


_lib = None


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype)) if arr is not None else None


def load():
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SplitPDeviceError(
            f"{LIB_PATH} not found: build it with `make -C splitp_amd/csrc` (or __graft_entry__.build()); "
            "splitp_amd has no CPU fallback"
        )
    # torch bundles its own libamdhip64.so.7; importing it first makes both use ONE HIP runtime
    # (same SONAME), which is what lets torch tensors and this library share device pointers.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    P = C.POINTER
    lib.sp_last_error.restype = C.c_char_p
    lib.sp_abi_version.restype = i32
    lib.sp_device_count.restype = i32
    sigs = {
        "sp_ctx_create": [i32, vp, P(vp)],
        "sp_ctx_destroy": [vp],
        "sp_ctx_set_stream": [vp, vp],
        "sp_ctx_set_option": [vp, C.c_char_p, i64],
        "sp_ctx_get_option": [vp, C.c_char_p, P(i64)],
        "sp_ctx_synchronize": [vp],
        "sp_ctx_set_gram_mode": [vp, i32],
        "sp_ctx_enable_timing": [vp, i32],
        "sp_ctx_reset_timing": [vp],
        "sp_ctx_phase_times": [vp, P(dbl), P(i64)],
        "sp_alignment_create": [vp, P(C.c_uint64), P(dbl), P(i64), i64, i32, i64, P(vp)],
        "sp_alignment_from_sequences": [vp, P(C.c_uint8), i32, i64, i64, P(vp)],
        "sp_alignment_from_site_keys": [vp, P(C.c_uint64), i64, i32, P(vp)],
        "sp_simulate_alignment": [vp, i32, P(C.c_int32), P(C.c_int32), P(dbl), i32, i64, C.c_uint64, P(vp)],
        "sp_alignment_destroy": [vp],
        "sp_alignment_info": [vp, P(i64), P(i32), P(i64), P(i32)],
        "sp_alignment_fetch": [vp, P(C.c_uint64), P(dbl), P(i64)],
        "sp_flatten_indices": [vp, P(C.c_int32), i32, P(C.c_int32), i32, P(i64), P(i64)],
        "sp_flatten_reduced_prepare": [vp, P(C.c_int32), i32, P(C.c_int32), i32, P(i64), P(i64)],
        "sp_flatten_reduced_fetch": [vp, P(dbl), P(i64), P(i64)],
        "sp_flatten_dense_counts": [vp, P(C.c_int32), i32, P(C.c_int32), i32, P(C.c_uint32)],
        "sp_subflatten": [vp, P(C.c_int32), i32, P(C.c_int32), i32, P(dbl)],
        "sp_moment_matrix": [vp, P(i64), P(dbl)],
        "sp_score_matrix_f64": [vp, P(dbl), i64, i64, i64, P(dbl)],
        "sp_score_coo_f64": [vp, P(i64), P(i64), P(dbl), i64, i64, i64, P(dbl)],
        "sp_divergence_matrix_f64": [vp, P(dbl), i64, i64, i64, P(dbl)],
        "sp_score_splits": [vp, P(C.c_int32), P(C.c_int32), i64, i32, P(dbl), vp, P(C.c_int32)],
        "sp_score_splits_async": [vp, P(C.c_int32), P(C.c_int32), i64, i32, vp, vp],
        "sp_score_splits_multi_async": [P(vp), i32, P(C.c_int32), P(C.c_int32), i64, vp, vp],
        "sp_score_all_splits": [vp, i32, i32, i32, P(i64), P(dbl), vp, P(C.c_int32)],
        "sp_score_all_splits_shard": [vp, i32, i32, i32, i32, i32, P(i64), P(dbl), vp, P(C.c_int32), vp],
        "sp_plan_create": [vp, i32, P(C.c_int32), P(C.c_int32), i64, P(vp)],
        "sp_plan_retain": [vp],
        "sp_plan_release": [vp],
        "sp_plan_info": [vp, P(i32), P(i64)],
        "sp_score_plan_async": [vp, P(vp), i32, vp, vp, vp],
        "sp_score_plan_steps": [vp, P(vp), i32, vp, i32, vp, i64, vp, i64],
        "sp_finish_flagged": [vp, P(C.c_int32), P(C.c_int32), i64, P(dbl), P(C.c_int32), P(i64)],
        "sp_debug_radix_sort": [vp, P(C.c_uint64), P(C.c_uint32), i32, i64, i64, C.c_uint, P(C.c_uint64), P(C.c_uint32)],
        "sp_node_create": [i32, P(vp)],
        "sp_node_destroy": [vp],
        "sp_node_info": [vp, P(i32)],
        "sp_node_score_all_splits": [vp, P(C.c_uint64), P(dbl), P(i64), i64, i32, i64, i32, i32, i32, P(i64), P(dbl),
                                     P(C.c_int32)],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    _lib = lib
    return lib


def check(status, allow_noconv=False):
    """Raise for a non-zero status.  allow_noconv: SP_ENOCONV (scores written, some flagged as upper estimates) is
    returned to the caller instead, who warns with the per-split status words in hand."""
    if status == SP_ENOCONV and allow_noconv:
        return status
    if status != 0:
        msg = load().sp_last_error().decode(errors="replace")
        codes = {1: ValueError, 4: NotImplementedError}
        raise codes.get(status, SplitPDeviceError)(f"libsplitp_hip: {msg} (status {status})")
    return 0


def device_count():
    return int(load().sp_device_count())


def require_gpu():
    if device_count() < 1:
        raise SplitPDeviceError("no HIP device visible; splitp_amd runs its hot path on the GPU only")
