"""Drop-in for the reference's Cython module ``teHmm._emission`` (_emission.pyx): canFast,
fastAllLogProbs, fastAccumulateStats, fastUpdateCounts with the same arguments and in-place semantics, computed by
libtehmm_hip.so."""
import ctypes

import numpy as np

from . import _lib
from ._lib import f64p, ptr


def _table_array(obs):
    from .track import TrackTable
    if isinstance(obs, TrackTable):
        obs = obs.getNumPyArray()
    return obs


def canFast(obs):
    """_emission.pyx:14-18."""
    from .track import TrackTable
    return isinstance(obs, TrackTable) or (
        isinstance(obs, np.ndarray) and obs.dtype in (np.int32, np.uint16, np.uint8))


def fastAllLogProbs(obs, logProbs, outProbs, normalize, segRatios):
    """_emission.pyx:20-48: outProbs [T,N] filled in place."""
    obs = _table_array(obs)
    assert isinstance(obs, np.ndarray) and isinstance(logProbs, np.ndarray)
    assert isinstance(outProbs, np.ndarray)
    assert obs.ndim == 2 and logProbs.ndim == 3
    assert logProbs.dtype == np.float64 and outProbs.dtype == np.float64
    assert outProbs.shape[0] == obs.shape[0] and logProbs.shape[0] == obs.shape[1]
    assert outProbs.flags.c_contiguous
    obs = np.ascontiguousarray(obs)
    lp = np.ascontiguousarray(logProbs)
    T, K = obs.shape
    _, N, S = lp.shape
    r = None if segRatios is None else np.ascontiguousarray(segRatios, dtype=np.float64)
    fn = {np.dtype(np.uint8): "tehmm_emission_u8", np.dtype(np.uint16): "tehmm_emission_u16",
          np.dtype(np.int32): "tehmm_emission_i32"}.get(obs.dtype)
    assert fn is not None, obs.dtype
    rc = getattr(_lib.load(), fn)(T, K, N, S, obs.ctypes.data_as(ctypes.c_void_p), ptr(lp, f64p),
                                  float(normalize), ptr(r, f64p), ptr(outProbs, f64p))
    _lib.check(rc, fn)


_SUFFIX = {np.dtype(np.uint8): "u8", np.dtype(np.uint16): "u16", np.dtype(np.int32): "i32"}


def fastAccumulateStats(obs, obsStats, posteriors, segRatios):
    """_emission.pyx:146-234: obsStats [K,N,S] += in place (uint8 / uint16 / int32 observations)."""
    obs = _table_array(obs)
    assert isinstance(obs, np.ndarray) and obs.ndim == 2
    assert isinstance(obsStats, np.ndarray) and obsStats.dtype == np.float64
    assert obsStats.flags.c_contiguous
    sfx = _SUFFIX.get(obs.dtype)
    assert sfx is not None, obs.dtype            # the reference: "assert False" (_emission.pyx:167)
    obs = np.ascontiguousarray(obs)
    T, K = obs.shape
    _, N, S = obsStats.shape
    post = np.ascontiguousarray(posteriors, dtype=np.float64)
    r = None if segRatios is None else np.ascontiguousarray(segRatios, dtype=np.float64)
    fn = "tehmm_accumulate_obs_" + sfx
    rc = getattr(_lib.load(), fn)(T, K, N, S, obs.ctypes.data_as(ctypes.c_void_p),
                                  ptr(obsStats, f64p), ptr(post, f64p), ptr(r, f64p))
    _lib.check(rc, fn)


def fastUpdateCountsBatch(intervals, trackTable, obsStats, segRatios):
    """All labelled intervals of ONE table in one device call: intervals = sequence of
    (chrom, start, end, state) in table-relative coordinates (TrackTable.getOverlapInTableCoords),
    processed in the given order -- the order emission.supervisedTrain feeds them to
    fastUpdateCounts one by one (emission.py:307-322)."""
    from ._lib import i32p, i64p
    obs = _table_array(trackTable)
    assert isinstance(obs, np.ndarray) and obs.ndim == 2
    assert isinstance(obsStats, np.ndarray) and obsStats.dtype == np.float64
    assert obsStats.flags.c_contiguous
    sfx = _SUFFIX.get(obs.dtype)
    assert sfx is not None, obs.dtype
    obs = np.ascontiguousarray(obs)
    T, K = obs.shape
    _, N, S = obsStats.shape
    n = len(intervals)
    if n == 0:
        return
    starts = np.ascontiguousarray([iv[1] for iv in intervals], dtype=np.int64)
    ends = np.ascontiguousarray([iv[2] for iv in intervals], dtype=np.int64)
    states = np.ascontiguousarray([iv[3] for iv in intervals], dtype=np.int32)
    r = None if segRatios is None else np.ascontiguousarray(segRatios, dtype=np.float64)
    fn = "tehmm_update_counts_" + sfx
    rc = getattr(_lib.load(), fn)(T, K, N, S, obs.ctypes.data_as(ctypes.c_void_p), n,
                                  ptr(starts, i64p), ptr(ends, i64p), ptr(states, i32p),
                                  ptr(r, f64p), ptr(obsStats, f64p))
    _lib.check(rc, fn)


def fastUpdateCounts(bedInterval, trackTable, obsStats, segRatios):
    """_emission.pyx:236-332: obsStats[track, state, obs[pos, track]] += 1 (or segRatios[pos]) for
    pos in [bedInterval[1], bedInterval[2]), state = bedInterval[3]."""
    from .track import TrackTable
    assert isinstance(trackTable, TrackTable)
    fastUpdateCountsBatch([bedInterval], trackTable, obsStats, segRatios)
