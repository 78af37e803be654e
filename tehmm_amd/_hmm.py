"""Drop-in for the reference's Cython module ``teHmm._hmm`` (_hmm.pyx): same function names,
argument order and in-place conventions, computed by libtehmm_hip.so on the MI355X.

Like the Cython typed-buffer arguments, arrays must be float64 (a ValueError is raised otherwise,
matching Cython's "Buffer dtype mismatch").
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import f64p, i64p, ptr


def _chk(a, ndim, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.ndim != ndim:
        raise ValueError("Buffer dtype mismatch or wrong number of dimensions for %s" % name)
    return np.ascontiguousarray(a)


def _ratios(segRatios, T):
    if segRatios is None:
        return None
    r = _chk(segRatios, 1, "segRatios")
    if r.shape[0] < T:
        raise ValueError("segRatios shorter than the observation count")
    return r


def _forward(n_observations, n_components, log_startprob, log_transmat, framelogprob, segRatios,
             fwdlattice):
    """_hmm.pyx:120-158.  fwdlattice is filled in place."""
    T, N = int(n_observations), int(n_components)
    pi = _chk(log_startprob, 1, "log_startprob")
    lt = _chk(log_transmat, 2, "log_transmat")
    fr = _chk(framelogprob, 2, "framelogprob")
    r = _ratios(segRatios, T)
    if not (isinstance(fwdlattice, np.ndarray) and fwdlattice.dtype == np.float64
            and fwdlattice.flags.c_contiguous):
        raise ValueError("fwdlattice must be a C-contiguous float64 array")
    _lib.check(_lib.load().tehmm_forward(T, N, ptr(pi, f64p), ptr(lt, f64p), ptr(fr, f64p),
                                         ptr(r, f64p), ptr(fwdlattice, f64p)), "tehmm_forward")


def _backward(n_observations, n_components, log_startprob, log_transmat, framelogprob, segRatios,
              bwdlattice):
    """_hmm.pyx:160-198.  bwdlattice is filled in place."""
    T, N = int(n_observations), int(n_components)
    pi = _chk(log_startprob, 1, "log_startprob")
    lt = _chk(log_transmat, 2, "log_transmat")
    fr = _chk(framelogprob, 2, "framelogprob")
    r = _ratios(segRatios, T)
    if not (isinstance(bwdlattice, np.ndarray) and bwdlattice.dtype == np.float64
            and bwdlattice.flags.c_contiguous):
        raise ValueError("bwdlattice must be a C-contiguous float64 array")
    _lib.check(_lib.load().tehmm_backward(T, N, ptr(pi, f64p), ptr(lt, f64p), ptr(fr, f64p),
                                          ptr(r, f64p), ptr(bwdlattice, f64p)), "tehmm_backward")


def _viterbi(n_observations, n_components, log_startprob, log_transmat, segRatios, framelogprob):
    """_hmm.pyx:201-259.  Returns (state_sequence int64[T], logprob)."""
    T, N = int(n_observations), int(n_components)
    pi = _chk(log_startprob, 1, "log_startprob")
    lt = _chk(log_transmat, 2, "log_transmat")
    fr = _chk(framelogprob, 2, "framelogprob")
    r = _ratios(segRatios, T)
    path = np.empty(T, dtype=np.int64)
    lp = ctypes.c_double(0.0)
    _lib.check(_lib.load().tehmm_viterbi(T, N, ptr(pi, f64p), ptr(lt, f64p), ptr(r, f64p),
                                         ptr(fr, f64p), ptr(path, i64p), ctypes.byref(lp)),
               "tehmm_viterbi")
    return path, lp.value


def _log_sum_lneta(n_observations, n_components, fwdlattice, log_transmat, bwdlattice,
                   framelogprob, logprob, segRatios, logsum_lneta):
    """_hmm.pyx:62-117.  logsum_lneta [N,N] is updated in place (caller zero-fills it)."""
    T, N = int(n_observations), int(n_components)
    f = _chk(fwdlattice, 2, "fwdlattice")
    lt = _chk(log_transmat, 2, "log_transmat")
    b = _chk(bwdlattice, 2, "bwdlattice")
    fr = _chk(framelogprob, 2, "framelogprob")
    r = _ratios(segRatios, T)
    if not (isinstance(logsum_lneta, np.ndarray) and logsum_lneta.dtype == np.float64
            and logsum_lneta.flags.c_contiguous):
        raise ValueError("logsum_lneta must be a C-contiguous float64 array")
    _lib.check(_lib.load().tehmm_xi_logsum(T, N, ptr(f, f64p), ptr(lt, f64p), ptr(b, f64p),
                                           ptr(fr, f64p), float(logprob), ptr(r, f64p),
                                           ptr(logsum_lneta, f64p)), "tehmm_xi_logsum")
