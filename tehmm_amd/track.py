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
        self.maskArray = None
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

    def getOverlapInTableCoords(self, bedInterval, startHint=None):
        """track.py:390-432: overlap of a genome-coordinate BED interval with this table, in TABLE
        coordinates (segment indices when the table is segmented), or None.  Binary search instead
        of the reference's linear scan from startHint (same result)."""
        assert len(bedInterval) > 2
        chrom, start, end = bedInterval[0], bedInterval[1], bedInterval[2]
        if not (self.chrom == chrom and self.start < end and self.end > start):
            return None
        overlap = [self.chrom, max(self.start, start), min(self.end, end)] + list(bedInterval[3:])
        if self.segOffsets is not None:
            offs = np.asarray(self.segOffsets, dtype=np.int64)
            genStart, genEnd = overlap[1] - self.start, overlap[2] - self.start
            if genStart < offs[0]:
                return None                                    # starts before the first segment
            first = int(np.searchsorted(offs, genStart, side="right")) - 1
            last = int(np.searchsorted(offs, genEnd, side="left")) - 1     # segment holding genEnd - 1
            if last < first:
                return None
            overlap[1], overlap[2] = first, last + 1
        else:
            overlap[1] -= self.start
            overlap[2] -= self.start
        return overlap

    def segment(self, segIntervals, trackList, interpolate=True):
        """track.py:449-495: keep one row per segment interval (chrom, start, end, ...).  Offsets (and
        the skipping of fully masked segments) are computed here; the O(T) work -- per-segment mode of
        the categorical tracks, mean of the gaussian ones, the gather -- is one device call
        (tehmm_segment_table_u8)."""
        import ctypes
        from . import _lib
        from ._lib import f64p, i64p, ptr, u8p
        first_end = self.origEnd if self.maskArray is not None else self.end
        ivs = [iv for iv in segIntervals if iv[0] == self.chrom and iv[1] >= self.start and iv[2] <= first_end]
        assert ivs and ivs[0][1] == self.start and ivs[-1][2] == first_end
        offs = np.asarray([int(iv[1]) - self.start for iv in ivs], dtype=np.int64)
        if self.maskArray is not None:
            run = self.getMaskRunningOffsets(reverseTransform=True)
            kept = self.maskArray[offs]                 # a segment is either all kept or all cut
            offs = (offs - run[offs])[kept]
        self.segOffsets = offs
        if len(offs) == 0:
            self.shape = (0, self.getNumTracks())
            return
        data = np.ascontiguousarray(self.data, dtype=np.uint8)
        T, K = data.shape
        is_g = np.zeros(K, dtype=np.uint8)
        mapback = np.zeros((K, 256), dtype=np.float64)
        if interpolate:
            for track in trackList:
                if track.getDist() == "gaussian":
                    is_g[track.getNumber()] = 1
                    mapback[track.getNumber()] = track.getValueMap().getMapBackTable(np.uint8)
        out = np.zeros((len(offs), K), dtype=np.uint8)
        means = np.zeros((len(offs), K), dtype=np.float64)
        if interpolate:
            _lib.check(_lib.load().tehmm_segment_table_u8(
                T, K, ptr(data, u8p), len(offs), ptr(offs, i64p), ptr(is_g, u8p), ptr(mapback, f64p),
                ptr(out, u8p), ptr(means, f64p)), "tehmm_segment_table_u8")
            for track in trackList:                     # O(segments): may create new symbols
                if track.getDist() == "gaussian":
                    k, vm = track.getNumber(), track.getValueMap()
                    out[:, k] = [vm.getMap(mv, update=True) for mv in means[:, k]]
        else:
            out = data[offs]
        del ctypes
        self.data = out
        self.shape = (len(self), self.getNumTracks())

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

    def setMaskTable(self, maskTable):
        """track.py:622-645: cut out every position a (binary) mask track covers; the flags, the running
        offsets (_track.runSum) and the compaction are one device call (tehmm_mask_table_u8)."""
        from . import _lib
        from ._lib import i32p, i64p, ptr, u8p
        self.maskArray = None
        if maskTable is None:
            return
        data = np.ascontiguousarray(self.data, dtype=np.uint8)
        mask = np.ascontiguousarray(maskTable.data, dtype=np.uint8)
        T, K = data.shape
        assert mask.shape[0] == T
        keep = np.zeros(T, dtype=np.uint8)
        run_full = np.zeros(T, dtype=np.int32)
        out = np.zeros((T, K), dtype=np.uint8)
        run_masked = np.zeros(T, dtype=np.int32)
        nk = np.zeros(1, dtype=np.int64)
        _lib.check(_lib.load().tehmm_mask_table_u8(T, K, ptr(data, u8p), mask.shape[1], ptr(mask, u8p),
                                                   ptr(keep, u8p), ptr(run_full, i32p), ptr(out, u8p),
                                                   ptr(run_masked, i32p), ptr(nk, i64p)), "tehmm_mask_table_u8")
        n = int(nk[0])
        self.maskArray = keep.astype(bool)
        self._run_full, self._run_masked = run_full, run_masked[:n].copy()
        self.data = out[:n].copy()
        self.shape = self.data.shape
        self.origEnd = self.end
        self.end = self.start + n

    def hasMask(self):
        return self.maskArray is not None

    def getMaskRunningOffsets(self, reverseTransform=False):
        """track.py:650-662: number of cut bases before every position (of the uncut table when
        reverseTransform, else of the kept rows)."""
        if not self.hasMask():
            return None
        return self._run_full if reverseTransform else self._run_masked


class CategoryMap(object):
    """Value <-> symbol dictionary of a track (track.py:668-799): symbols are handed out in order of
    first appearance starting at `reserved`; numeric tracks bin their values first
    (key = str(int(scale * (value + shift))) or str(int(log_base(value + shift))))."""

    def __init__(self, reserved=1, defaultVal=None, scale=None, logScale=None, shift=None):
        self.fwd, self.back = {}, {}
        self.reserved = reserved
        self.scale = None if logScale is not None else scale
        self.logBase = logScale
        self.shift = None if shift is None else float(shift)
        self.defaultVal = defaultVal
        self.missingVal = max(0, reserved - 1)
        if defaultVal is not None:
            self.missingVal = int(self.getMap(defaultVal, update=True))

    def _key(self, x):
        y = x if self.shift is None else float(x) + self.shift
        if self.scale is not None:
            return str(int(self.scale * float(y)))
        if self.logBase is not None:
            return str(int(np.log(float(y)) / np.log(self.logBase)))
        return y

    def _unkey(self, key):
        y = key
        if self.scale is not None:
            y = float(key) / float(self.scale)
        elif self.logBase is not None:
            y = float(self.logBase) ** float(key)
        return y if self.shift is None else float(y) - self.shift

    def update(self, value):
        key = self._key(value)
        if key not in self.fwd:
            sym = len(self.fwd) + self.reserved
            self.fwd[key], self.back[sym] = sym, key

    def has(self, value):
        return self._key(value) in self.fwd

    def getMap(self, value, update=False):
        key = self._key(value)
        if update and key is not None and key not in self.fwd:
            self.update(value)
        return self.fwd.get(key, self.missingVal)

    def getMapBack(self, sym):
        if sym in self.back:
            return self._unkey(self.back[sym])
        if self.defaultVal is not None:
            return self._unkey(self.back[self.getMap(self.defaultVal)])
        return None

    def getMapBackTable(self, dtype):
        table = np.full(int(np.iinfo(dtype).max) + 1, float(np.iinfo(np.int64).max), dtype=np.float64)
        for sym in range(len(table)):
            v = self.getMapBack(sym)
            if v is not None:
                table[sym] = v
        return table

    def getMissingVal(self):
        return self.missingVal

    def getReserved(self):
        return self.reserved

    def __len__(self):
        return len(self.fwd) + max(0, self.reserved - 1)

    def sort(self):
        """Re-number so that symbols ascend with the (numeric where possible) keys."""
        keys = list(self.fwd)
        try:
            order = sorted(keys, key=float)
        except (TypeError, ValueError):
            order = sorted(keys)
        self.fwd = {k: i + self.reserved for i, k in enumerate(order)}
        self.back = {v: k for k, v in self.fwd.items()}


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
