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

    def estep_device(self, batch, use_ratios, stats):
        """E-step with the raw statistics ADDED into the device buffer `stats` (DeviceStats)."""
        lp = ctypes.c_double(0.0)
        _lib.check(_lib.load().tehmm_estep_batch_device(self._h, batch._h, int(bool(use_ratios)), stats.ptr,
                                                        ctypes.byref(lp)), "tehmm_estep_batch_device")
        return lp.value

    def mstep(self, stats, do_start, do_trans, do_emission, start_prior, trans_prior, fudge, gauss=None):
        """MultitrackHmm._do_mstep on the device; gauss = (track indices, values [n][S], uniform mix) or
        None.  Returns the gaussian (mu, sigma) table [n][N][2] or None."""
        n_g, gt, gv, mix, gp = 0, None, None, 0.1, None
        if gauss is not None and do_emission:     # (the gaussian refit is part of the emission update only)
            gt = np.ascontiguousarray(gauss[0], dtype=np.int32)
            gv = np.ascontiguousarray(gauss[1], dtype=np.float64)
            mix = float(gauss[2])
            n_g = len(gt)
            assert gv.shape == (n_g, self.S)
            gp = np.zeros((n_g, self.N, 2), dtype=np.float64)
        _lib.check(_lib.load().tehmm_model_mstep(self._h, stats.ptr, int(bool(do_start)), int(bool(do_trans)),
                                                 int(bool(do_emission)), float(start_prior), float(trans_prior),
                                                 float(fudge), n_g, ptr(gt, i32p), ptr(gv, f64p), mix,
                                                 ptr(gp, f64p)), "tehmm_model_mstep")
        return gp

    def get_params(self, log_probs_like):
        """(log_transmat [N,N], log_startprob [N], logProbs [K,N,S]) of the handle; the padding cells of
        logProbs are taken from `log_probs_like`."""
        lt = np.zeros((self.N, self.N), dtype=np.float64)
        pi = np.zeros(self.N, dtype=np.float64)
        lp = np.ascontiguousarray(log_probs_like, dtype=np.float64).copy()
        _lib.check(_lib.load().tehmm_model_get_params(self._h, ptr(lt, f64p), ptr(pi, f64p), ptr(lp, f64p)),
                   "tehmm_model_get_params")
        return lt, pi, lp

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


class DeviceStats(object):
    """The flat fp64 buffer of raw E-step statistics on the device (include/tehmm_hip.h,
    "device-resident Baum-Welch").  With torch.distributed initialised on the nccl backend the buffer
    is a torch CUDA tensor, so that RCCL all-reduces it in place; otherwise the library allocates it."""

    def __init__(self, model):
        self.model = model
        self.size = int(_lib.load().tehmm_model_stats_size(model._h))
        self.tensor = None
        self._own = None
        t = _dist_tensor(self.size)
        if t is not None:
            self.tensor = t
            self.ptr = ctypes.c_void_p(t.data_ptr())
        else:
            p = vp()
            _lib.check(_lib.load().tehmm_stats_alloc(model._h, ctypes.byref(p)), "tehmm_stats_alloc")
            self._own = p
            self.ptr = p

    def zero(self):
        _lib.check(_lib.load().tehmm_stats_zero(self.model._h, self.ptr), "tehmm_stats_zero")

    def to_host(self):
        out = np.zeros(self.size, dtype=np.float64)
        _lib.check(_lib.load().tehmm_stats_copy(self.model._h, self.ptr, ptr(out, f64p), 0), "tehmm_stats_copy")
        return out

    def from_host(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        assert buf.shape == (self.size,)
        _lib.check(_lib.load().tehmm_stats_copy(self.model._h, self.ptr, ptr(buf, f64p), 1), "tehmm_stats_copy")

    def head(self):
        lp, n = ctypes.c_double(0.0), ctypes.c_double(0.0)
        _lib.check(_lib.load().tehmm_stats_head(self.ptr, ctypes.byref(lp), ctypes.byref(n)), "tehmm_stats_head")
        return lp.value, n.value

    def close(self):
        if self._own is not None:
            _lib.load().tehmm_stats_free(self._own)
            self._own = None
        self.tensor = None

    __del__ = close


def _dist_tensor(size):
    """A zeroed CUDA tensor when this process is a rank of an nccl (= RCCL) process group, else None."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return None
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"):
        return None
    return torch.zeros(size, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))


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
            # uint8 rows as they are; uint16 / int32 tables (the reference's other two IntegerTrackTable types) through
            # tehmm_batch_create_u16 / _i32, which refuse symbols beyond 255
            obs = np.asarray(obs)
            kind = {np.dtype(np.uint16): "u16", np.dtype(np.int32): "i32"}.get(obs.dtype)
            obs = np.ascontiguousarray(obs, dtype=obs.dtype if kind else np.uint8)
            assert obs.ndim == 2 and obs.shape[0] == offsets[-1]
            self.K = obs.shape[1]
            o = obs.ctypes.data_as(ctypes.c_void_p)
            rr = None if ratios is None else np.ascontiguousarray(ratios, dtype=np.float64)
            if rr is not None:
                assert rr.shape[0] == offsets[-1]
            r = None if rr is None else rr.ctypes.data_as(ctypes.c_void_p)
            if kind:
                fn = getattr(_lib.load(), "tehmm_batch_create_" + kind)
                _lib.check(fn(self.n, ptr(offsets, i64p), self.K, o, r, ctypes.byref(h)), "tehmm_batch_create_" + kind)
                self._h = h
                self.N = None
                self.total = int(offsets[-1])
                return
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

    def reset_cache(self):
        """Forget what earlier evaluations derived from the observations (tehmm_batch_reset_cache)."""
        _lib.check(_lib.load().tehmm_batch_reset_cache(self._h), "tehmm_batch_reset_cache")

    def paths(self, row0=0, row1=None, pinned=True):
        """pinned: the result lives in pinned host memory (one DMA at link speed; the block returns to the
        library's pool when the array dies)."""
        row1 = self.total if row1 is None else row1
        big = pinned and (row1 - row0) >= (1 << 19)
        out = _lib.pinned_empty(row1 - row0, np.int64) if big else np.empty(row1 - row0, dtype=np.int64)
        _lib.check(_lib.load().tehmm_batch_get_paths(self._h, row0, row1, ptr(out, i64p)),
                   "tehmm_batch_get_paths")
        return out

    def posteriors(self, n_states=None, row0=0, row1=None, pinned=True):
        n_states = self.N if n_states is None else n_states
        row1 = self.total if row1 is None else row1
        big = pinned and (row1 - row0) * n_states >= (1 << 19)
        out = (_lib.pinned_empty((row1 - row0, n_states), np.float64) if big
               else np.empty((row1 - row0, n_states), dtype=np.float64))
        _lib.check(_lib.load().tehmm_batch_get_posteriors(self._h, row0, row1, ptr(out, f64p)),
                   "tehmm_batch_get_posteriors")
        return out

    def interval_logprobs(self):
        """Per-interval forward log-likelihoods of the last estep / posterior evaluation."""
        out = np.zeros(self.n, dtype=np.float64)
        _lib.check(_lib.load().tehmm_batch_get_interval_logprobs(self._h, ptr(out, f64p)),
                   "tehmm_batch_get_interval_logprobs")
        return out

    def posterior_masksum(self, mask, row0=0, row1=None):
        """sum_j posteriors[r, j] * mask[j] for rows [row0, row1) (teHmmEval.py:270-272), reduced on
        the device: 8 instead of 8 N bytes per row cross PCIe."""
        row1 = self.total if row1 is None else row1
        mask = np.ascontiguousarray(mask, dtype=np.float64)
        assert mask.shape[0] == self.N
        out = np.empty(row1 - row0, dtype=np.float64)
        _lib.check(_lib.load().tehmm_batch_posterior_masksum(self._h, ptr(mask, f64p), row0, row1,
                                                             ptr(out, f64p)),
                   "tehmm_batch_posterior_masksum")
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


def eval_stream(model, obs, offsets, ratios=None, group_rows=4_000_000, viterbi=True, posterior=True,
                mask=None, use_ratios=True):
    """teHmmEval over host-resident intervals with the result transfer hidden behind the evaluation: the intervals
    are cut into groups of about `group_rows` positions, a worker thread creates and evaluates group g + 1 (H2D of the
    observations, tehmm_eval_batch) while this thread fetches group g's paths and posteriors over PCIe into pinned
    host memory -- the reference writes its per-chromosome outputs one after the other as well
    (bin/teHmmEval.py:312-383).  With the full posterior rows the link is the bound (8 N bytes per position against
    ~1.4e9 positions/s on the device), so the job takes the transfer time plus one group's evaluation instead of
    their sum.  mask: fetch the masked posterior sums (teHmmEval.py:270-272) instead of the rows.

    Returns (paths list, posteriors-or-masked-sums list, viterbi_logprob, forward_logprob) in interval order; the
    per-group arrays are views of pinned blocks."""
    import threading
    import queue
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    groups, g0 = [], 0
    while g0 < n:
        g1 = g0 + 1
        while g1 < n and offsets[g1 + 1] - offsets[g0] <= group_rows:
            g1 += 1
        groups.append((g0, g1))
        g0 = g1
    ready = queue.Queue(maxsize=2)          # at most two evaluated groups wait for their transfer

    def producer():
        try:
            for (a, b) in groups:
                r0, r1 = int(offsets[a]), int(offsets[b])
                hb = HipBatch(obs[r0:r1], offsets[a:b + 1] - offsets[a], None if ratios is None else ratios[r0:r1])
                res = model.eval(hb, viterbi=viterbi, posterior=posterior, use_ratios=use_ratios and ratios is not None)
                ready.put((a, b, hb, res, None))
        except BaseException as exc:        # (handed to the consumer: it re-raises)
            ready.put((None, None, None, None, exc))

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    paths, posts = [None] * n, [None] * n
    vlp = np.zeros(n) if viterbi else None
    flp = np.zeros(n) if posterior else None
    for _ in groups:
        a, b, hb, res, exc = ready.get()
        if exc is not None:
            th.join()
            raise exc
        lo = offsets[a:b + 1] - offsets[a]
        if viterbi:
            p = hb.paths()
            vlp[a:b] = res["viterbi_logprob"]
            for i in range(a, b):
                paths[i] = p[int(lo[i - a]):int(lo[i - a + 1])]
        if posterior:
            q = hb.posterior_masksum(mask) if mask is not None else hb.posteriors()
            flp[a:b] = res["forward_logprob"]
            for i in range(a, b):
                posts[i] = q[int(lo[i - a]):int(lo[i - a + 1])]
        hb.close()
    th.join()
    return paths, posts, vlp, flp
