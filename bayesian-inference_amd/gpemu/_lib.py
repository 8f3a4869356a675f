"""ctypes binding of libgpemu.so (include/gpemu.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C bayesian-inference_amd/csrc``
and lives next to this file.  There is no CPU implementation: if the shared object is missing, or
no HIP device is visible, every compute call raises ``GpemuError`` -- loudly, never a silent
fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPEMU_LIBRARY: another build of the library (A/B measurements of one kernel variant against the in-tree build)
LIB_PATH = os.environ.get("GPEMU_LIBRARY") or os.path.join(_HERE, "libgpemu.so")

c_double_p = C.POINTER(C.c_double)
c_i64 = C.c_int64


class GpemuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgpemu error {code}: {msg}")
        self.code = code


_lib = None

# name -> (restype, argtypes); mirrors include/gpemu.h one to one
_SIGNATURES = {
    "gpemu_version": (C.c_char_p, []),
    "gpemu_last_error": (C.c_char_p, []),
    "gpemu_device_count": (C.c_int, []),
    "gpemu_device_name": (C.c_int, [C.c_int, C.c_char_p, c_i64]),
    "gpemu_device_bus_id": (C.c_int, [C.c_int, C.c_char_p, c_i64]),
    "gpemu_device_memory": (C.c_int, [C.c_int, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "gpemu_model_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, c_i64, c_i64, c_i64, c_i64,
                                     C.c_int, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 10),
    "gpemu_model_destroy": (C.c_int, [C.c_void_p]),
    "gpemu_model_dims": (C.c_int, [C.c_void_p] + [C.POINTER(c_i64)] * 4),
    "gpemu_model_device": (C.c_int, [C.c_void_p]),
    "gpemu_model_sync": (C.c_int, [C.c_void_p]),
    "gpemu_model_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "gpemu_model_profile_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_gp_predict": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_gp_predict_dev": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_predict_full": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "gpemu_predict_full_dev": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_double, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "gpemu_likelihood_setup": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_double, c_i64, C.c_void_p]),
    "gpemu_likelihood_setup_chains": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_double, c_i64, C.c_void_p]),
    "gpemu_logpost": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_int]),
    "gpemu_logpost_dev": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "gpemu_fit_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, c_i64, c_i64, C.c_void_p, C.c_int, C.c_double,
                                   C.c_int, C.c_int, C.c_double]),
    "gpemu_fit_destroy": (C.c_int, [C.c_void_p]),
    "gpemu_fit_lml": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.POINTER(C.c_double), C.c_void_p]),
    "gpemu_fit_lml_batch": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_fit_factor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_double)]),
    "gpemu_kernel_matrix": (C.c_int, [C.c_int, c_i64, c_i64, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_double,
                                      C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "gpemu_cholesky": (C.c_int, [C.c_int, c_i64, C.c_void_p]),
    "gpemu_pca_fit": (C.c_int, [C.c_int, c_i64, c_i64, C.c_void_p, c_i64] + [C.c_void_p] * 10),
    "gpemu_truncation_cov": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_sampler_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, c_i64,
                                       C.c_double, C.c_uint64]),
    "gpemu_sampler_create_chains": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, c_i64,
                                              C.c_double, C.c_void_p, C.c_int]),
    "gpemu_sampler_destroy": (C.c_int, [C.c_void_p]),
    "gpemu_sampler_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gpemu_sampler_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_sampler_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_sampler_reset": (C.c_int, [C.c_void_p]),
    "gpemu_sampler_run": (C.c_int, [C.c_void_p, c_i64, C.c_int]),
    "gpemu_sampler_step_host_rng": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int]),
    "gpemu_sampler_get_chain": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p]),
    "gpemu_sampler_get_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "gpemu_sampler_acf": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, C.c_void_p]),
    "gpemu_sampler_reserve_chain": (C.c_int, [C.c_void_p, c_i64]),
    "gpemu_sampler_begin_step": (C.c_int, [C.c_void_p]),
    "gpemu_sampler_half_propose_eval": (C.c_int, [C.c_void_p, C.c_int, c_i64, c_i64, C.c_void_p]),
    "gpemu_sampler_half_accept": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "gpemu_sampler_end_step": (C.c_int, [C.c_void_p, C.c_int]),
    "gpemu_sampler_check": (C.c_int, [C.c_void_p]),
    "gpemu_comm_unique_id": (C.c_int, [C.c_char_p, C.c_void_p]),
    "gpemu_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p]),
    "gpemu_comm_destroy": (C.c_int, [C.c_void_p]),
    "gpemu_comm_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpemu_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p]),
    "gpemu_sampler_run_sharded": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_int]),
    "gpemu_sampler_peer_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gpemu_sampler_peer_import": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gpemu_sampler_run_peer": (C.c_int, [C.c_void_p, c_i64, C.c_int]),
    "gpemu_sampler_peer_selftest": (C.c_int, [C.c_void_p]),
    "gpemu_sampler_peer_share": (C.c_int, [C.c_void_p, C.c_int]),
    "gpemu_sampler_snapshot": (C.c_int, [C.c_void_p]),
    "gpemu_sampler_restore": (C.c_int, [C.c_void_p]),
    "gpemu_halfstep_small_launches": (C.c_int64, []),
    "gpemu_philox4x32": (C.c_int, [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If libgpemu
    pulled in the system copy first, a later ``import torch`` would bring a second HIP runtime into the
    process and see no GPUs.  Loading torch's copy first (by path, without importing torch) makes the
    loader satisfy libgpemu's libamdhip64.so.7 dependency with it, so both share one runtime."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # fall back to the system ROCm runtime


def lib():
    """Load libgpemu.so once; raise if it has not been built."""
    global _lib
    if _lib is None:
        # Sharing device memory across processes (the peer transport of the walker-sharded run: hipIpcGetMemHandle /
        # hipIpcOpenMemHandle of the exchange buffers, and RCCL itself) needs dmabuf IPC on hosts whose driver has no
        # legacy IPC: the runtime reads this when it initialises, i.e. before the first HIP call of the process.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise GpemuError(-100, f"{LIB_PATH} not found: build it with "
                                   "`python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise GpemuError(code, lib().gpemu_last_error().decode("utf-8", "replace"))


def last_error() -> str:
    return lib().gpemu_last_error().decode("utf-8", "replace")


def device_count() -> int:
    return int(lib().gpemu_device_count())


def default_device():
    """Device of this process: one process per GPU under torch.distributed.run (LOCAL_RANK), else device 0.
    GPEMU_DEVICE overrides.  Taken modulo the number of visible devices."""
    n = require_device()
    idx = os.environ.get("GPEMU_DEVICE", os.environ.get("LOCAL_RANK", "0"))
    try:
        return int(idx) % n
    except ValueError:
        return 0


def resolve_device(device):
    return default_device() if device is None else int(device)


def require_device():
    n = device_count()
    if n <= 0:
        raise GpemuError(-3, "no HIP device visible; libgpemu has no CPU implementation")
    return n


def device_free_bytes(device=None):
    """Free device memory in bytes (hipMemGetInfo), or None where it cannot be asked (no device: the caller's own
    limits apply and the compute call that follows fails loudly)."""
    try:
        free, total = c_i64(0), c_i64(0)
        if lib().gpemu_device_memory(int(resolve_device(device)), C.byref(free), C.byref(total)) != 0:
            return None
        return int(free.value)
    except Exception:
        return None


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
