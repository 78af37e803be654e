"""The step after the path: what bin/teHmmEval.py writes per row (teHmmEval.py:216-275).

The reference formats one text line per table row in the interpreter (statesToBed), which dominates its
wall-clock at 100 Mb.  Here the coordinates of every row (segment lengths, mask offsets) and the
masked posterior sum are reduced on the device, and the lines are formatted by the library's native
writer (tehmm_write_bed).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import f64p, i32p, i64p, ptr


def bedCoords(trackTable):
    """(starts, ends) int64 arrays, one entry per row of the table (teHmmEval.py:251-266)."""
    n = len(trackTable)
    offs = trackTable.getSegmentOffsets()
    offs = None if offs is None else np.ascontiguousarray(offs, dtype=np.int64)
    mo = trackTable.getMaskRunningOffsets() if hasattr(trackTable, "getMaskRunningOffsets") else None
    mo = None if mo is None else np.ascontiguousarray(mo, dtype=np.int32)
    starts, ends = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    _lib.check(_lib.load().tehmm_bed_coords(n, int(trackTable.getStart()), int(trackTable.getEnd()),
                                            ptr(offs, i64p), ptr(mo, i32p), 0 if mo is None else len(mo),
                                            ptr(starts, i64p), ptr(ends, i64p)), "tehmm_bed_coords")
    return starts, ends


def _write(path, append, chrom, starts, ends, states=None, names=None, values=None):
    n = len(starts)
    st = None if states is None else np.ascontiguousarray(states, dtype=np.int64)
    vals = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    arr, n_names = None, 0
    if names is not None:
        n_names = len(names)
        arr = (ctypes.c_char_p * n_names)(*[str(x).encode() for x in names])
    _lib.check(_lib.load().tehmm_write_bed(str(path).encode(), 1 if append else 0, str(chrom).encode(), n,
                                           ptr(starts, i64p), ptr(ends, i64p), ptr(st, i64p), n_names, arr,
                                           ptr(vals, f64p)), "tehmm_write_bed")


def statesToBed(trackTable, states, bedPath=None, posteriorSums=None, posteriorsPath=None, stateNames=None,
                append=True):
    """teHmmEval.py:238-275.  states: integer state per row; posteriorSums: the masked posterior sum per
    row (HipBatch.posterior_masksum) -- written one row late, wrapping, as the reference does (quirk Q15:
    row i carries posteriors[i - 1])."""
    starts, ends = bedCoords(trackTable)
    chrom = trackTable.getChrom()
    if bedPath is not None:
        _write(bedPath, append, chrom, starts, ends, states=states, names=stateNames)
    if posteriorSums is not None and posteriorsPath is not None:
        _write(posteriorsPath, append, chrom, starts, ends, values=np.roll(np.asarray(posteriorSums), 1))


def bic(model, totalScore, totalDatapoints):
    """Bayesian information criterion as teHmmEval.py:216-234 writes it: -2 lnL + k (ln n + ln 2 pi), with
    lnL the summed Viterbi score, n = rows x tracks and k = model.getNumFreeParameters() (0 when the
    model cannot say: the reference swallows the exception).  Returns (bic, k)."""
    lnL = float(totalScore)
    try:
        k = float(model.getNumFreeParameters())
    except Exception:
        k = 0.0
    n = float(totalDatapoints)
    return -2.0 * lnL + k * (np.log(n) + np.log(2 * np.pi)), k


def writeBic(path, model, totalScore, totalDatapoints):
    """The two-line --bic file of teHmmEval.py:216-234."""
    value, k = bic(model, totalScore, totalDatapoints)
    em = model.getEmissionModel()
    with open(path, "w") as f:
        f.write("%f\n" % value)
        f.write("# = -2.0 * lnL + k * (lnN + ln(2 * np.pi))\n"
                "# where lnL=%f  k=%d (%d states)  N=%d (%d obs * %d tracks)  lnN=%f\n" % (
                    float(totalScore), int(k), em.getNumStates(), int(totalDatapoints),
                    totalDatapoints / em.getNumTracks(), em.getNumTracks(), np.log(float(totalDatapoints))))
    return value
