"""ctypes binding of the CPU oracle (oracle/tehmm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (tehmm_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtehmm_oracle.so")
_lib = None

_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "tehmm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_logsumexp.restype = ctypes.c_double
    return _lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _ratios(r):
    if r is None or (hasattr(r, "__len__") and len(r) == 0):
        return None
    return np.ascontiguousarray(r, dtype=np.float64)


def emission(obs, log_probs, normalize=1.0, ratios=None):
    obs = np.ascontiguousarray(obs)
    T, K = obs.shape
    lp = _f64(log_probs)
    _, N, S = lp.shape
    out = np.zeros((T, N))
    r = _ratios(ratios)
    fn = {np.dtype(np.uint8): "oracle_emission_u8", np.dtype(np.uint16): "oracle_emission_u16",
          np.dtype(np.int32): "oracle_emission_i32"}[obs.dtype]
    getattr(lib(), fn)(ctypes.c_int64(T), K, N, S, obs.ctypes.data_as(ctypes.c_void_p),
                       _p(lp, _f64p), ctypes.c_double(normalize), _p(r, _f64p), _p(out, _f64p))
    return out


def forward(pi, lt, frame, ratios=None):
    frame = _f64(frame)
    T, N = frame.shape
    out = np.zeros((T, N))
    r = _ratios(ratios)
    lib().oracle_forward(ctypes.c_int64(T), N, _p(_f64(pi), _f64p), _p(_f64(lt), _f64p),
                         _p(frame, _f64p), _p(r, _f64p), _p(out, _f64p))
    return out


def backward(pi, lt, frame, ratios=None):
    frame = _f64(frame)
    T, N = frame.shape
    out = np.zeros((T, N))
    r = _ratios(ratios)
    lib().oracle_backward(ctypes.c_int64(T), N, _p(_f64(pi), _f64p), _p(_f64(lt), _f64p),
                          _p(frame, _f64p), _p(r, _f64p), _p(out, _f64p))
    return out


def viterbi(pi, lt, frame, ratios=None):
    frame = _f64(frame)
    T, N = frame.shape
    path = np.zeros(T, dtype=np.int64)
    lp = ctypes.c_double(0.0)
    r = _ratios(ratios)
    lib().oracle_viterbi(ctypes.c_int64(T), N, _p(_f64(pi), _f64p), _p(_f64(lt), _f64p),
                         _p(r, _f64p), _p(frame, _f64p), _p(path, _i64p), ctypes.byref(lp))
    return path, lp.value


def logsumexp(x):
    x = _f64(x)
    return lib().oracle_logsumexp(len(x), _p(x, _f64p))


def xi_logsum(fwd, lt, bwd, frame, logprob, ratios=None):
    fwd, bwd, frame = _f64(fwd), _f64(bwd), _f64(frame)
    T, N = frame.shape
    out = np.zeros((N, N))
    r = _ratios(ratios)
    lib().oracle_xi_logsum(ctypes.c_int64(T), N, _p(fwd, _f64p), _p(_f64(lt), _f64p), _p(bwd, _f64p),
                           _p(frame, _f64p), ctypes.c_double(logprob), _p(r, _f64p), _p(out, _f64p))
    return out


def accumulate_obs(obs, obs_stats, post, ratios=None):
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    T, K = obs.shape
    assert obs_stats.dtype == np.float64 and obs_stats.flags.c_contiguous
    _, N, S = obs_stats.shape
    r = _ratios(ratios)
    lib().oracle_accumulate_obs_u8(ctypes.c_int64(T), K, N, S, _p(obs, _u8p), _p(obs_stats, _f64p),
                                   _p(_f64(post), _f64p), _p(r, _f64p))
    return obs_stats


def posteriors(fwd, bwd, add_eps):
    fwd, bwd = _f64(fwd), _f64(bwd)
    T, N = fwd.shape
    out = np.zeros((T, N))
    lib().oracle_posteriors(ctypes.c_int64(T), N, _p(fwd, _f64p), _p(bwd, _f64p), int(add_eps),
                            _p(out, _f64p))
    return out


def decode(obs, log_probs, pi, lt, normalize=1.0, ratios=None):
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    T, K = obs.shape
    lp = _f64(log_probs)
    _, N, S = lp.shape
    path = np.zeros(T, dtype=np.int64)
    out = ctypes.c_double(0.0)
    r = _ratios(ratios)
    rc = lib().oracle_decode(ctypes.c_int64(T), K, N, S, _p(obs, _u8p), _p(lp, _f64p),
                             ctypes.c_double(normalize), _p(_f64(pi), _f64p), _p(_f64(lt), _f64p),
                             _p(r, _f64p), _p(path, _i64p), ctypes.byref(out))
    assert rc == 0
    return out.value, path


def score_samples(obs, log_probs, pi, lt, normalize=1.0):
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    T, K = obs.shape
    lp = _f64(log_probs)
    _, N, S = lp.shape
    post = np.zeros((T, N))
    out = ctypes.c_double(0.0)
    rc = lib().oracle_score_samples(ctypes.c_int64(T), K, N, S, _p(obs, _u8p), _p(lp, _f64p),
                                    ctypes.c_double(normalize), _p(_f64(pi), _f64p),
                                    _p(_f64(lt), _f64p), ctypes.byref(out), _p(post, _f64p))
    assert rc == 0
    return out.value, post


def estep(seqs, log_probs, pi, lt, normalize=1.0, ratios_list=None):
    """Sum of the per-sequence E-step statistics (basehmm.py:504-523, hmm.py:545-574)."""
    lp = _f64(log_probs)
    K, N, S = lp.shape
    start = np.zeros(N)
    trans = np.zeros((N, N))
    obs_stats = np.zeros((K, N, S))
    logprob = ctypes.c_double(0.0)
    pi, lt = _f64(pi), _f64(lt)
    for i, obs in enumerate(seqs):
        obs = np.ascontiguousarray(obs, dtype=np.uint8)
        r = _ratios(ratios_list[i]) if ratios_list is not None else None
        rc = lib().oracle_estep_seq(ctypes.c_int64(obs.shape[0]), K, N, S, _p(obs, _u8p), _p(lp, _f64p),
                                    ctypes.c_double(normalize), _p(pi, _f64p), _p(lt, _f64p),
                                    _p(r, _f64p), _p(start, _f64p), _p(trans, _f64p),
                                    _p(obs_stats, _f64p), ctypes.byref(logprob))
        assert rc == 0
    return dict(start=start, trans=trans, obs=obs_stats, logprob=logprob.value, nobs=len(seqs))


def eval_batch(obs, offsets, log_probs, pi, lt, normalize=1.0, ratios=None, want_post=True,
               n_threads=1):
    """Reference teHmmEval flow (score_samples + decode per interval) over a batch, threaded
    over intervals like teHmmEval.py --chroms/--proc (bin/teHmmEval.py:312-383)."""
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    lp = _f64(log_probs)
    K, N, S = lp.shape
    total = int(offsets[-1])
    paths = np.zeros(total, dtype=np.int64)
    n = len(offsets) - 1
    vlp = np.zeros(n)
    flp = np.zeros(n)
    post = np.zeros((total, N)) if want_post else None
    r = _ratios(ratios)
    rc = lib().oracle_eval_batch(n, _p(offsets, _i64p), K, N, S, _p(obs, _u8p), _p(lp, _f64p),
                                 ctypes.c_double(normalize), _p(_f64(pi), _f64p), _p(_f64(lt), _f64p),
                                 _p(r, _f64p), _p(paths, _i64p), _p(vlp, _f64p), _p(flp, _f64p),
                                 _p(post, _f64p), int(n_threads))
    assert rc == 0
    return paths, vlp, flp, post
