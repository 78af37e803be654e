"""Constants and log helpers shared by the host-side mirror of teHmm's model API.

Mirrors the parts of the reference's ``common.py`` that carry path arithmetic:
``LOGZERO``/``EPSILON``/``myLog`` (common.py:24-33).  Zero probabilities become
``-1e100`` (not ``-inf``) in the log-transition and log-start tables (quirk Q1).
"""
import logging

import numpy as np

LOGZERO = -1e100                      # common.py:24
EPSILON = np.finfo(float).eps         # common.py:25
ZEROLOGPROB = -1e200                  # basehmm.py:64, _hmm.pyx:60
NEGINF = -np.inf                      # basehmm.py:66
F32_EPS = float(np.finfo(np.float32).eps)   # basehmm.py:271

logger = logging.getLogger("teHmm")


def myLog(x, logZeroVal=LOGZERO, epsilonVal=EPSILON):
    """Vectorised ``log`` that maps |x| < eps to ``logZeroVal`` (common.py:27-33)."""
    a = np.asarray(x, dtype=np.float64)
    small = np.abs(a) < epsilonVal
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.log(np.where(small, 1.0, a))
    out = np.where(small, logZeroVal, out)
    if out.ndim == 0:
        return float(out)
    return out


def logsumexp(arr, axis=0):
    """log(sum(exp(arr))) along ``axis`` exactly as basehmm.py:70-93 does it."""
    arr = np.rollaxis(np.asarray(arr, dtype=np.float64), axis)
    vmax = arr.max(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.log(np.sum(np.exp(arr - vmax), axis=0))
    out += vmax
    return out


def normalize(A, axis=None):
    """basehmm.py:113-141: adds machine eps, then divides by the sum (in place add)."""
    A += np.finfo(float).eps
    Asum = A.sum(axis)
    if axis and A.ndim > 1:
        Asum[Asum == 0] = 1
        shape = list(A.shape)
        shape[axis] = 1
        Asum.shape = shape
    return A / Asum
