"""ctypes binding of libppca_hip.so (the C-ABI of include/ppca_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
present, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPCA_HIP_LIB", os.path.join(_HERE, "libppca_hip.so"))  # override: diagnostic builds

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_void_pp = C.POINTER(C.c_void_p)


class PPCAError(RuntimeError):
    """Raised for any non-zero ppca_status (the reference panics / raises)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[ppca_hip {code}] {message}")
        self.code = code


class Prior(C.Structure):
    _fields_ = [
        ("has_mean_prior", C.c_int32),
        ("mean", c_double_p),
        ("mean_covariance", c_double_p),
        ("has_isotropic_noise_prior", C.c_int32),
        ("isotropic_noise_alpha", C.c_double),
        ("isotropic_noise_beta", C.c_double),
        ("transformation_precision", C.c_double),
    ]


class SynthSpec(C.Structure):
    _fields_ = [
        ("row_offset", C.c_int64),
        ("n_rows", C.c_int64),
        ("d", C.c_int32),
        ("k", C.c_int32),
        ("sigma", C.c_double),
        ("mask_prob", C.c_double),
        ("mask_kind", C.c_int32),
        ("mask_run", C.c_int32),
        ("seed", C.c_uint64),
        ("transform", c_double_p),
        ("mean", c_double_p),
    ]


# name -> (restype, argtypes); every symbol include/ppca_hip.h declares
SIGNATURES = {
    "ppca_last_error": (C.c_char_p, []),
    "ppca_abi_version": (C.c_int32, []),
    "ppca_path_kind": (C.c_int32, [C.c_int32, C.c_int32]),
    "ppca_ctx_create": (C.c_int, [C.c_int32, C.c_void_p, c_void_pp]),
    "ppca_ctx_destroy": (C.c_int, [C.c_void_p]),
    "ppca_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ppca_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "ppca_ctx_trim": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ppca_ctx_enable_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "ppca_ctx_kernel_time": (C.c_int, [C.c_void_p, c_double_p, C.POINTER(C.c_int64), C.c_int32]),
    "ppca_dataset_from_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, c_void_pp]),
    "ppca_dataset_from_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, c_void_pp]),
    "ppca_dataset_generate": (C.c_int, [C.c_void_p, C.POINTER(SynthSpec), c_void_pp]),
    "ppca_dataset_with_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_void_pp]),
    "ppca_dataset_slice": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_void_pp]),
    "ppca_dataset_concat": (C.c_int, [C.c_void_p, c_void_pp, C.c_int32, c_void_pp]),
    "ppca_dataset_free": (C.c_int, [C.c_void_p]),
    "ppca_dataset_len": (C.c_int64, [C.c_void_p]),
    "ppca_dataset_output_size": (C.c_int32, [C.c_void_p]),
    "ppca_dataset_device_x": (C.c_void_p, [C.c_void_p]),
    "ppca_dataset_device_weights": (C.c_void_p, [C.c_void_p]),
    "ppca_dataset_to_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ppca_dataset_weights_to_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ppca_dataset_empty_dimensions": (C.c_int, [C.c_void_p, c_int32_p]),
    "ppca_model_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, c_void_pp]),
    "ppca_model_alloc": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_void_pp]),
    "ppca_model_download": (C.c_int, [C.c_void_p, c_double_p, C.c_void_p, C.c_void_p]),
    "ppca_model_free": (C.c_int, [C.c_void_p]),
    "ppca_model_output_size": (C.c_int32, [C.c_void_p]),
    "ppca_model_state_size": (C.c_int32, [C.c_void_p]),
    "ppca_stats_len": (C.c_int64, [C.c_int32, C.c_int32]),
    "ppca_em_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppca_em_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Prior), C.c_void_p]),
    "ppca_em_finalize_host": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Prior), c_double_p, C.c_void_p, C.c_void_p]),
    "ppca_em_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Prior), C.c_void_p, c_double_p]),
    "ppca_em_last_llk": (C.c_int, [C.c_void_p, c_double_p]),
    "ppca_stats_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppca_llk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_double_p, C.c_void_p]),
    "ppca_llks_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppca_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppca_reconstruct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_void_pp]),
    "ppca_covariance_diagonal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, c_void_pp]),
    "ppca_mix_em_step": (C.c_int, [C.c_void_p, C.c_void_p, c_void_pp, C.c_void_p, C.c_int32, C.POINTER(Prior), c_void_pp, C.c_void_p, c_double_p]),
    "ppca_mix_last_rows_used": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "ppca_mix_llk": (C.c_int, [C.c_void_p, C.c_void_p, c_void_pp, C.c_void_p, C.c_int32, c_double_p, C.c_void_p, C.c_void_p]),
    "ppca_mix_responsibilities_dev": (C.c_int, [C.c_void_p, C.c_void_p, c_void_pp, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ppca_mix_component_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, c_double_p, C.POINTER(C.c_int64)]),
    "ppca_vector_max_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_double_p]),
    "ppca_vector_exp_shift_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p]),
    "ppca_vector_sum_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_double_p]),
    "ppca_mix_reconstruct": (C.c_int, [C.c_void_p, C.c_void_p, c_void_pp, C.c_void_p, C.c_int32, C.c_int32, c_void_pp]),
    "ppca_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ppca_comm_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, c_void_pp]),
    "ppca_comm_create_all": (C.c_int, [c_void_pp, C.c_int32, c_void_pp]),
    "ppca_comm_destroy": (C.c_int, [C.c_void_p]),
    "ppca_comm_n_ranks": (C.c_int32, [C.c_void_p]),
    "ppca_comm_rank": (C.c_int32, [C.c_void_p]),
    "ppca_comm_backend": (C.c_char_p, []),
    "ppca_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ppca_em_step_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Prior), C.c_void_p, c_double_p]),
    "ppca_mix_em_step_sharded": (C.c_int, [C.c_void_p, C.c_void_p, c_void_pp, C.c_void_p, C.c_int32, C.POINTER(Prior), c_void_pp, C.c_void_p, c_double_p]),
    "ppca_em_step_group": (C.c_int, [c_void_pp, C.c_int32, c_void_pp, c_void_pp, C.POINTER(Prior), c_void_pp, c_double_p]),
    "ppca_ctx_set_grid_limit": (C.c_int, [C.c_void_p, C.c_int32]),
    "ppca_ctx_set_heavy_rows": (C.c_int, [C.c_void_p, C.c_int32]),
    "ppca_debug_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "ppca_em_last_guard": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    "ppca_dataset_scale_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_double]),
    "ppca_em_last_fallback": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, C.POINTER(C.c_int64), c_double_p]),
    "ppca_gram_engine": (C.c_int, [C.c_void_p, C.c_void_p, c_int32_p]),
    "ppca_debug_mfma_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ppca_debug_mfma_i8_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def _share_torch_hip_runtime() -> None:
    """A PyTorch-ROCm wheel bundles its own libamdhip64 / libhsa-runtime64 / librccl.  If this library were bound to
    /opt/rocm's runtime and torch were imported later, the process would hold TWO HIP runtimes: torch streams handed
    to ppca_ctx_set_stream, and torch's RCCL, would belong to the other one.  So when a torch wheel with a bundled
    runtime is installed, that runtime is loaded first (by path, without importing torch) and libppca_hip.so binds
    to it through the common soname -- the configuration bench.py always ran in.  PPCA_SYSTEM_HIP=1 opts out."""
    import sys

    if "torch" in sys.modules or os.environ.get("PPCA_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def lib():
    """The loaded shared library (raises ImportError loudly when it was not built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} is missing: build the HIP extension with `python -m ppca_rs_amd.build` "
                        "(there is no CPU fallback)"
                    )
                _share_torch_hip_runtime()
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().ppca_last_error()
        raise PPCAError(rc, msg.decode() if msg else "unknown error")


def f64(x, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One GPU + one stream (ppca_ctx)."""

    def __init__(self, device: int = -1, stream: int | None = None):
        h = C.c_void_p()
        check(lib().ppca_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h

    def set_stream(self, stream: int | None) -> None:
        check(lib().ppca_ctx_set_stream(self.handle, C.c_void_p(stream) if stream else None))

    def synchronize(self) -> None:
        check(lib().ppca_ctx_synchronize(self.handle))

    def trim(self) -> int:
        """Returns the device blocks the context keeps for reuse to the device; the number of bytes released."""
        got = C.c_int64(0)
        check(lib().ppca_ctx_trim(self.handle, C.byref(got)))
        return int(got.value)

    def set_grid_limit(self, n_workgroups: int = 0) -> None:
        """Test hook (ppca_ctx_set_grid_limit): cap the workgroups of every persistent-grid launch; 0 restores."""
        check(lib().ppca_ctx_set_grid_limit(self.handle, int(n_workgroups)))

    def set_heavy_rows(self, max_rows: int = 8) -> None:
        """Test hook (ppca_ctx_set_heavy_rows): rows of a tile that may go round the fixed-point form of the EM pass's mask-side
        statistics (default 8; 0: every tile with such a row raises its workgroup's exponents, as before ABI 6)."""
        check(lib().ppca_ctx_set_heavy_rows(self.handle, int(max_rows)))

    def debug_counters(self, reset: bool = True):
        """ppca_debug_counters: 8 counters of the EM pass's int8 statistics contraction (see include/ppca_hip.h)."""
        out = (C.c_int64 * 8)()
        check(lib().ppca_debug_counters(self.handle, out, int(reset)))
        return [int(v) for v in out]

    def last_guard(self):
        """ppca_em_last_guard: (gram_unsafe, stats_unsafe) of the most recent fused EM pass on this context."""
        g, w = C.c_int32(0), C.c_int32(0)
        check(lib().ppca_em_last_guard(self.handle, C.byref(g), C.byref(w)))
        return int(g.value), int(w.value)

    def last_fallback(self):
        """ppca_em_last_fallback: (mode, workgroups recomputed, rows recomputed, ms of the second stages since the last call) of the
        most recent fused EM pass on this context: mode 0 nothing, 1 the whole pass on the fp64 engine, 2 the flagged slices only."""
        m, g, r, ms = C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_double(0.0)
        check(lib().ppca_em_last_fallback(self.handle, C.byref(m), C.byref(g), C.byref(r), C.byref(ms)))
        return int(m.value), int(g.value), int(r.value), float(ms.value)

    def enable_timing(self, on: bool) -> None:
        check(lib().ppca_ctx_enable_timing(self.handle, int(on)))

    def kernel_time(self, reset: bool = True):
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        check(lib().ppca_ctx_kernel_time(self.handle, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.ppca_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    """Process-wide context on LOCAL_RANK's GPU (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("LOCAL_RANK", os.environ.get("PPCA_DEVICE", "0")))
        _default_ctx = Context(dev)
    return _default_ctx


def set_default_context(ctx: Context) -> None:
    global _default_ctx
    _default_ctx = ctx
