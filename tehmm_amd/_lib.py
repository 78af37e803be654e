"""ctypes loader of libtehmm_hip.so (the C ABI in include/tehmm_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEHMM_HIP_LIB") or os.path.join(_HERE, "libtehmm_hip.so")

f64p = ctypes.POINTER(ctypes.c_double)
i64p = ctypes.POINTER(ctypes.c_int64)
i32p = ctypes.POINTER(ctypes.c_int32)
u8p = ctypes.POINTER(ctypes.c_uint8)
vp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double

# every symbol include/tehmm_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "tehmm_abi_version": (c_int, []),
    "tehmm_last_error": (ctypes.c_char_p, []),
    "tehmm_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "tehmm_set_device": (c_int, [c_int]),
    "tehmm_max_states": (c_int, []),
    "tehmm_emission_u8": (c_int, [c_i64, c_int, c_int, c_int, vp, f64p, c_dbl, f64p, f64p]),
    "tehmm_emission_u16": (c_int, [c_i64, c_int, c_int, c_int, vp, f64p, c_dbl, f64p, f64p]),
    "tehmm_emission_i32": (c_int, [c_i64, c_int, c_int, c_int, vp, f64p, c_dbl, f64p, f64p]),
    "tehmm_forward": (c_int, [c_i64, c_int, f64p, f64p, f64p, f64p, f64p]),
    "tehmm_backward": (c_int, [c_i64, c_int, f64p, f64p, f64p, f64p, f64p]),
    "tehmm_viterbi": (c_int, [c_i64, c_int, f64p, f64p, f64p, f64p, i64p, f64p]),
    "tehmm_xi_logsum": (c_int, [c_i64, c_int, f64p, f64p, f64p, f64p, c_dbl, f64p, f64p]),
    "tehmm_accumulate_obs_u8": (c_int, [c_i64, c_int, c_int, c_int, vp, f64p, f64p, f64p]),
    "tehmm_accumulate_obs_u16": (c_int, [c_i64, c_int, c_int, c_int, vp, f64p, f64p, f64p]),
    "tehmm_accumulate_obs_i32": (c_int, [c_i64, c_int, c_int, c_int, vp, f64p, f64p, f64p]),
    "tehmm_update_counts_u8": (c_int, [c_i64, c_int, c_int, c_int, vp, c_int, i64p, i64p, i32p, f64p, f64p]),
    "tehmm_update_counts_u16": (c_int, [c_i64, c_int, c_int, c_int, vp, c_int, i64p, i64p, i32p, f64p, f64p]),
    "tehmm_update_counts_i32": (c_int, [c_i64, c_int, c_int, c_int, vp, c_int, i64p, i64p, i32p, f64p, f64p]),
    "tehmm_model_stats_size": (c_i64, [vp]),
    "tehmm_stats_alloc": (c_int, [vp, ctypes.POINTER(vp)]),
    "tehmm_stats_zero": (c_int, [vp, vp]),
    "tehmm_stats_free": (c_int, [vp]),
    "tehmm_stats_head": (c_int, [vp, f64p, f64p]),
    "tehmm_stats_copy": (c_int, [vp, vp, f64p, c_int]),
    "tehmm_estep_batch_device": (c_int, [vp, vp, c_int, vp, f64p]),
    "tehmm_model_mstep": (c_int, [vp, vp, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_int, i32p, f64p, c_dbl,
                                  f64p]),
    "tehmm_model_get_params": (c_int, [vp, f64p, f64p, f64p]),
    "tehmm_segment_table_u8": (c_int, [c_i64, c_int, u8p, c_i64, i64p, u8p, f64p, u8p, f64p]),
    "tehmm_mask_table_u8": (c_int, [c_i64, c_int, u8p, c_int, u8p, u8p, i32p, u8p, i32p, i64p]),
    "tehmm_batch_get_interval_logprobs": (c_int, [vp, f64p]),
    "tehmm_batch_posterior_masksum": (c_int, [vp, f64p, c_i64, c_i64, f64p]),
    "tehmm_bed_coords": (c_int, [c_i64, c_i64, c_i64, i64p, i32p, c_i64, i64p, i64p]),
    "tehmm_write_bed": (c_int, [ctypes.c_char_p, c_int, ctypes.c_char_p, c_i64, i64p, i64p, i64p, c_int,
                                ctypes.POINTER(ctypes.c_char_p), f64p]),
    "tehmm_model_create": (c_int, [c_int, c_int, c_int, f64p, f64p, f64p, c_dbl, i32p,
                                   ctypes.POINTER(vp)]),
    "tehmm_model_destroy": (c_int, [vp]),
    "tehmm_batch_create": (c_int, [c_int, i64p, c_int, vp, vp, c_int, ctypes.POINTER(vp)]),
    "tehmm_batch_create_u16": (c_int, [c_int, i64p, c_int, vp, vp, ctypes.POINTER(vp)]),
    "tehmm_batch_create_i32": (c_int, [c_int, i64p, c_int, vp, vp, ctypes.POINTER(vp)]),
    "tehmm_batch_destroy": (c_int, [vp]),
    "tehmm_batch_total": (c_i64, [vp]),
    "tehmm_batch_reset_cache": (c_int, [vp]),
    "tehmm_eval_batch": (c_int, [vp, vp, c_int, f64p, f64p]),
    "tehmm_batch_get_paths": (c_int, [vp, c_i64, c_i64, i64p]),
    "tehmm_batch_get_posteriors": (c_int, [vp, c_i64, c_i64, f64p]),
    "tehmm_host_alloc": (c_int, [ctypes.c_size_t, ctypes.POINTER(vp)]),
    "tehmm_host_free": (c_int, [vp]),
    "tehmm_trim_pools": (c_int, []),
    "tehmm_batch_device_ptrs": (c_int, [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]),
    "tehmm_estep_batch": (c_int, [vp, vp, c_int, f64p, f64p, f64p, f64p]),
    "tehmm_batch_last_timing": (c_int, [vp, c_int, ctypes.POINTER(ctypes.c_char_p), f64p]),
    "tehmm_debug_read_stamps": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_int]),
}

EVAL_VITERBI = 1
EVAL_POSTERIOR = 2
EVAL_USE_RATIOS = 4

_lib = None


class TeHmmHipError(RuntimeError):
    code = 0                      # the TEHMM_ERR_* code of the failed call (-3: TEHMM_ERR_UNSUPPORTED)


def load():
    """Load the shared library (no GPU needed just to load it and resolve the symbols)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TeHmmHipError(
                "%s not found: build it with `python -m tehmm_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback for the teHmm hot path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().tehmm_last_error()
        err = TeHmmHipError("%s failed (%d): %s" % (what or "tehmm call", rc,
                                                    msg.decode() if msg else "?"))
        err.code = int(rc)
        raise err


def device_count():
    n = c_int(0)
    rc = load().tehmm_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


class _PinnedBlock(object):
    """Owner of one tehmm_host_alloc block; the numpy arrays made over it keep it alive."""

    def __init__(self, nbytes):
        p = vp()
        check(load().tehmm_host_alloc(max(int(nbytes), 1), ctypes.byref(p)), "tehmm_host_alloc")
        self.ptr = p

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.tehmm_host_free(self.ptr)
            self.ptr = None


def trim_pools():
    """Return the library's cached device blocks and pinned host blocks to the system."""
    check(load().tehmm_trim_pools(), "tehmm_trim_pools")


def pinned_empty(shape, dtype):
    """numpy array over pinned host memory (tehmm_host_alloc): D2H copies into it are one DMA.  If the pinned
    allocation fails (a multi-GB request on a host short of lockable memory) the array is ordinary pageable memory:
    slower transfers, same results."""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if shape else 1
    try:
        blk = _PinnedBlock(n * dt.itemsize)
    except TeHmmHipError:
        return np.empty(shape, dtype=dt)
    buf = (ctypes.c_char * max(n * dt.itemsize, 1)).from_address(blk.ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    return _PinnedArray(arr, blk)


class _PinnedArray(np.ndarray):
    """ndarray over a pinned block.  Only the array made by pinned_empty owns the block; views keep it alive through
    .base, and results COMPUTED from it (ufunc outputs, reductions, copies) are plain ndarrays in ordinary memory --
    a small result must not keep a multi-GB pinned block alive."""

    def __new__(cls, arr, blk):
        obj = arr.view(cls)
        obj._blk = blk
        return obj

    def __array_finalize__(self, obj):
        self._blk = None

    def __array_wrap__(self, out, context=None, return_scalar=False):
        out = np.asarray(out)
        return out[()] if return_scalar else out


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)
