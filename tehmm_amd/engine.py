"""Device-resident handles over the fused C-ABI entry points (include/tehmm_hip.h).

HipModel  = the model tables teHmm keeps in MultitrackHmm / its emission model, on the GPU.
HipBatch  = the observation columns of many independent intervals (TrackTables), on the GPU,
            plus the result buffers of the last evaluation.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (EVAL_POSTERIOR, EVAL_USE_RATIOS, EVAL_VITERBI, f64p, i32p, i64p, ptr, vp)


class HipModel(object):
    def __init__(self, log_transmat, log_startprob, log_probs, normalize=1.0,
                 symbols_per_track=None):
        lt = _lib.as_f64(log_transmat)
        pi = _lib.as_f64(log_startprob)
        lp = _lib.as_f64(log_probs)
        assert lt.ndim == 2 and lt.shape[0] == lt.shape[1] == pi.shape[0] == lp.shape[1]
        self.N = lt.shape[0]
        self.K, _, self.S = lp.shape
        spt = None
        if symbols_per_track is not None:
            spt = np.ascontiguousarray(symbols_per_track, dtype=np.int32)
            assert spt.shape[0] == self.K
        h = vp()
        _lib.check(_lib.load().tehmm_model_create(self.N, self.K, self.S, ptr(lt, f64p), ptr(pi, f64p),
                                                  ptr(lp, f64p), float(normalize), ptr(spt, i32p),
                                                  ctypes.byref(h)), "tehmm_model_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().tehmm_model_destroy(self._h)
            self._h = None

    __del__ = close

    def eval(self, batch, viterbi=True, posterior=True, use_ratios=True):
        """Run BaseHMM.decode and/or BaseHMM.score_samples over every interval of ``batch``.
        Returns per-interval log-probabilities; paths / posteriors stay on the device until
        fetched with batch.paths() / batch.posteriors()."""
        flags = (EVAL_VITERBI if viterbi else 0) | (EVAL_POSTERIOR if posterior else 0)
        if use_ratios:
            flags |= EVAL_USE_RATIOS
        vlp = np.zeros(batch.n) if viterbi else None
        flp = np.zeros(batch.n) if posterior else None
        _lib.check(_lib.load().tehmm_eval_batch(self._h, batch._h, flags, ptr(vlp, f64p), ptr(flp, f64p)),
                   "tehmm_eval_batch")
        batch.N = self.N
        return {"viterbi_logprob": vlp, "forward_logprob": flp}

    def estep(self, batch, use_ratios, start, trans, obs_stats):
        """Accumulate Baum-Welch sufficient statistics of every interval into the given arrays;
        returns the summed forward log-likelihood."""
        for a in (start, trans, obs_stats):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        lp = ctypes.c_double(0.0)
        _lib.check(_lib.load().tehmm_estep_batch(self._h, batch._h, int(bool(use_ratios)),
                                                 ptr(start, f64p), ptr(trans, f64p),
                                                 ptr(obs_stats, f64p), ctypes.byref(lp)),
                   "tehmm_estep_batch")
        return lp.value


class HipBatch(object):
    def __init__(self, obs, offsets, ratios=None, device_ptrs=False, K=None):
        """obs: uint8 [total, K] concatenated interval rows (host array), offsets: int64 [n+1].
        device_ptrs=True: ``obs`` / ``ratios`` are integer device addresses (e.g. a torch
        tensor's data_ptr()) and K must be given."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n = len(offsets) - 1
        self.offsets = offsets
        h = vp()
        if device_ptrs:
            assert K is not None
            self.K = int(K)
            o = ctypes.c_void_p(int(obs))
            r = ctypes.c_void_p(int(ratios)) if ratios else None
        else:
            obs = np.ascontiguousarray(obs, dtype=np.uint8)
            assert obs.ndim == 2 and obs.shape[0] == offsets[-1]
            self.K = obs.shape[1]
            o = obs.ctypes.data_as(ctypes.c_void_p)
            rr = None if ratios is None else np.ascontiguousarray(ratios, dtype=np.float64)
            if rr is not None:
                assert rr.shape[0] == offsets[-1]
            r = None if rr is None else rr.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.load().tehmm_batch_create(self.n, ptr(offsets, i64p), self.K, o, r,
                                                  1 if device_ptrs else 0, ctypes.byref(h)),
                   "tehmm_batch_create")
        self._h = h
        self.N = None
        self.total = int(offsets[-1])

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().tehmm_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def paths(self, row0=0, row1=None):
        row1 = self.total if row1 is None else row1
        out = np.empty(row1 - row0, dtype=np.int64)
        _lib.check(_lib.load().tehmm_batch_get_paths(self._h, row0, row1, ptr(out, i64p)),
                   "tehmm_batch_get_paths")
        return out

    def posteriors(self, n_states=None, row0=0, row1=None):
        n_states = self.N if n_states is None else n_states
        row1 = self.total if row1 is None else row1
        out = np.empty((row1 - row0, n_states), dtype=np.float64)
        _lib.check(_lib.load().tehmm_batch_get_posteriors(self._h, row0, row1, ptr(out, f64p)),
                   "tehmm_batch_get_posteriors")
        return out

    def device_ptrs(self):
        p, q = vp(), vp()
        _lib.check(_lib.load().tehmm_batch_device_ptrs(self._h, ctypes.byref(p), ctypes.byref(q)),
                   "tehmm_batch_device_ptrs")
        return p.value, q.value

    def timing(self):
        names = (ctypes.c_char_p * 16)()
        ms = (ctypes.c_double * 16)()
        n = _lib.load().tehmm_batch_last_timing(self._h, 16, names, ms)
        return {names[i].decode(): ms[i] for i in range(n)}
