"""The data contract the hot path consumes: TrackTable / IntegerTrackTable (track.py:352-662).

Only what the HMM path touches is mirrored: a uint8 [T, K] column-major-by-position array
(``getNumPyArray``), optional segment offsets and ``getSegmentLengthsAsRatio``.  Loading tracks
from BED / XML files is outside the hot path ("next" rows).
"""
import numpy as np

INTEGER_ARRAY_TYPE = np.uint8   # track.py:23


class TrackTable(object):
    def __init__(self, numTracks, chrom, start, end):
        assert end > start
        self.numTracks = numTracks
        self.chrom = chrom
        self.start = start
        self.end = end
        self.origEnd = end
        self.segOffsets = None
        self.shape = (len(self), self.getNumTracks())

    def __len__(self):
        if self.segOffsets is None:
            return self.end - self.start
        return len(self.segOffsets)

    def getNumTracks(self):
        return self.numTracks

    def getChrom(self):
        return self.chrom

    def getStart(self):
        return self.start

    def getEnd(self):
        return self.end

    def getNumPyArray(self):
        raise RuntimeError("Not implemented")

    def getSegmentOffsets(self):
        return self.segOffsets

    def setSegmentOffsets(self, segOffsets):
        """Attach segment offsets to an already-compressed table (one row per segment)."""
        self.segOffsets = None if segOffsets is None else np.asarray(segOffsets, dtype=np.int64)
        self.shape = (len(self), self.getNumTracks())

    def getSegmentLength(self, i):
        """track.py:497-502."""
        if i == len(self.segOffsets) - 1:
            return self.end - (self.start + self.segOffsets[-1])
        elif i < len(self.segOffsets) - 1:
            return self.segOffsets[i + 1] - self.segOffsets[i]

    def getSegmentLengthsAsRatio(self, effectiveSegmentLength):
        """track.py:504-513, vectorised: segment length / effective segment length."""
        if self.segOffsets is None:
            return None
        eff = float(effectiveSegmentLength)
        assert eff >= 1
        offs = np.asarray(self.segOffsets, dtype=np.int64)
        ends = np.concatenate([offs[1:], [self.end - self.start]])
        return (ends - offs).astype(np.float64) / eff


class IntegerTrackTable(TrackTable):
    def __init__(self, numTracks, chrom, start, end, dtype=INTEGER_ARRAY_TYPE):
        super(IntegerTrackTable, self).__init__(numTracks, chrom, start, end)
        self.data = np.zeros((end - start, numTracks), dtype=dtype)
        self.iinfo = np.iinfo(dtype)
        self.maskArray = None

    def __getitem__(self, index):
        return self.data[index]

    def writeRow(self, row, rowArray):
        """Write one track (row) of values, clamping to the dtype range (track.py:563-580)."""
        assert row < self.getNumTracks()
        assert len(rowArray) == len(self)
        self.data[:, row] = np.clip(np.asarray(rowArray), self.iinfo.min, self.iinfo.max)

    def getNumPyArray(self):
        return self.data

    def getRow(self, row):
        return self.data[:, row]

    def initRow(self, row, val):
        self.data[:, row] = val

    def setData(self, data):
        data = np.ascontiguousarray(data, dtype=self.data.dtype)
        assert data.ndim == 2 and data.shape[1] == self.numTracks
        self.data = data
        return self


class IdentityValueMap(object):
    """Value map of a numeric (binned) track: symbol s (1-based, 0 = missing) <-> value s - 1.
    Stands in for CategoryMap (track.py:668-760) where only getMapBack is needed (the gaussian
    re-fit of the M-step, emission.py:520-545)."""

    def __init__(self, scale=1.0, shift=0.0):
        self.scale = float(scale)
        self.shift = float(shift)

    def getMapBack(self, symbol):
        return (float(symbol) - 1.0) * self.scale + self.shift

    def getMap(self, value, update=False):
        return int(round((float(value) - self.shift) / self.scale)) + 1


class Track(object):
    """The per-track metadata the hot path's callers look at (track.py:27-99)."""

    def __init__(self, name, number, dist="multinomial", valueMap=None):
        self.name = name
        self.number = number
        self.dist = dist
        self.valueMap = valueMap if valueMap is not None else IdentityValueMap()

    def getName(self):
        return self.name

    def getNumber(self):
        return self.number

    def getDist(self):
        return self.dist

    def getValueMap(self):
        return self.valueMap


class TrackList(list):
    def getTrackByNumber(self, n):
        return self[n]

    def getTrackByName(self, name):
        for t in self:
            if t.getName() == name:
                return t
        return None


class TrackData(object):
    """Container of the TrackTables of a set of intervals (track.py:838-977 without the BED / XML
    loading, which stays outside the hot path)."""

    def __init__(self, trackTableList=None, trackList=None, numSymbolsPerTrack=None):
        self.trackTableList = list(trackTableList) if trackTableList is not None else []
        self.trackList = trackList
        self.numSymbolsPerTrack = numSymbolsPerTrack

    def getTrackTableList(self):
        return self.trackTableList

    def getTrackList(self):
        return self.trackList

    def getNumTracks(self):
        return self.trackTableList[0].getNumTracks() if self.trackTableList else 0

    def getNumTrackTables(self):
        return len(self.trackTableList)

    def getNumSymbolsPerTrack(self):
        return self.numSymbolsPerTrack
