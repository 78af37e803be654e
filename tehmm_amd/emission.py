"""Emission models of the multitrack HMM -- host-side mirror of the reference's ``emission.py``.

The table ``logProbs[track, state, symbol]`` (symbol 0 = "missing data", log-prob 0) is the only
thing the device kernels consume; everything that touches T-sized data (allLogProbs,
accumulateStats) goes through ``tehmm_amd._emission`` (the HIP library), the O(model) M-step stays
in NumPy exactly as the reference does it.
"""
import copy
import itertools
from functools import reduce

import numpy as np
from numpy.testing import assert_array_almost_equal
from scipy import stats

from ._emission import canFast, fastAccumulateStats, fastAllLogProbs, fastUpdateCountsBatch
from .common import EPSILON, logger, myLog, normalize
from .track import TrackTable


class IndependentMultinomialEmissionModel(object):
    """emission.py:30-481.  Tracks are independent: P(obs | state) = prod_k P_k(obs_k | state)."""

    def __init__(self, numStates, numSymbolsPerTrack, params=None, zeroAsMissingData=True,
                 fudge=0.0, normalizeFac=0.0, randomize=False, effectiveSegmentLength=None,
                 random_state=None, randRange=(0.1, 0.9), uniformMixProb=0.1):
        self.numStates = numStates
        self.numTracks = len(numSymbolsPerTrack)
        self.numSymbolsPerTrack = list(numSymbolsPerTrack)
        self.random_state = random_state
        if self.random_state is None:
            self.random_state = np.random.mtrand._rand
        self.logProbs = None
        self.zeroAsMissingData = zeroAsMissingData
        self.fudge = fudge
        self.normalizeFac = 1.
        if normalizeFac > 0:
            self.normalizeFac = float(normalizeFac) / float(self.numTracks)     # emission.py:58-60
        self.effectiveSegmentLength = effectiveSegmentLength
        self.randRange = float(randRange[0]), float(randRange[1])
        self.uniformMixProb = float(uniformMixProb)
        self.initParams(params=params, randomize=randomize)

    # ---- accessors
    def getLogProbs(self):
        return self.logProbs

    def getNumStates(self):
        return self.numStates

    def getNumTracks(self):
        return self.numTracks

    def getNumSymbolsPerTrack(self):
        return self.numSymbolsPerTrack

    def getTrackSymbols(self, track):
        offset = 1 if self.zeroAsMissingData is True else 0
        for i in range(offset, self.numSymbolsPerTrack[track] + offset):
            yield i

    def getSymbols(self):
        if self.numTracks == 1:
            for i in self.getTrackSymbols(0):
                yield [i]
        else:
            valArrays = []
            for track in range(self.numTracks):
                if self.numSymbolsPerTrack[track] > 0:
                    valArrays.append([x for x in self.getTrackSymbols(track)])
                else:
                    valArrays.append([0])
            for val in itertools.product(*valArrays):
                yield val

    def _randDist(self, numPoints):
        samples = self.random_state.random_sample(numPoints)
        samples = self.randRange[0] + samples * (self.randRange[1] - self.randRange[0])
        return normalize(samples)

    def initParams(self, params=None, randomize=False):
        """emission.py:128-170: flat (or random, or given) distributions; a leading 1 for symbol 0."""
        offset = 1 if self.zeroAsMissingData else 0
        self.logProbs = np.zeros((self.numTracks, self.numStates, offset + max(self.numSymbolsPerTrack)),
                                 dtype=np.float64)
        for i in range(self.numTracks):
            for j in range(self.numStates):
                if params is None:
                    if randomize is False:
                        dist = normalize(1. + np.zeros(self.numSymbolsPerTrack[i], dtype=np.float64))
                    else:
                        dist = normalize(self._randDist(self.numSymbolsPerTrack[i]))
                else:
                    dist = np.array(params[i][j], dtype=np.float64)
                if self.zeroAsMissingData is True:
                    dist = np.append([1.], dist)
                with np.errstate(divide="ignore"):
                    self.logProbs[i, j, :len(dist)] = np.log(dist)
        self.validate()

    def singleLogProb(self, state, singleObs):
        logProb = 0.0
        for track, obsSymbol in enumerate(singleObs):
            logProb += self.logProbs[track][state][int(obsSymbol)]
        return logProb * self.normalizeFac

    def allLogProbs(self, obs):
        """emission.py:179-198: [T, numStates] frame; segment ratios only for TrackTables."""
        obsLogProbs = np.zeros((obs.shape[0], self.numStates), dtype=np.float64)
        segRatios = self.getSegmentRatios(obs)
        if canFast(obs):
            fastAllLogProbs(obs, self.logProbs, obsLogProbs, self.normalizeFac, segRatios)
        else:
            arr = np.asarray(obs)
            for i in range(len(arr)):
                for state in range(self.numStates):
                    obsLogProbs[i, state] = self.singleLogProb(state, arr[i])
                    if segRatios is not None:
                        obsLogProbs[i, state] *= segRatios[i]
        return obsLogProbs

    def initStats(self):
        """emission.py:208-219."""
        obsStats = np.zeros((self.numTracks, self.numStates, np.max(self.numSymbolsPerTrack) + 1),
                            dtype=np.float64)
        for track in range(self.numTracks):
            obsStats[track, :, :self.numSymbolsPerTrack[track] + 1] += self.fudge
        return obsStats

    def accumulateStats(self, obs, obsStats, posteriors):
        """emission.py:221-241."""
        assert obs.shape[1] == self.numTracks
        segRatios = self.getSegmentRatios(obs)
        if canFast(obs):
            fastAccumulateStats(obs, obsStats, posteriors, segRatios)
        else:
            arr = np.asarray(obs)
            for i in range(len(arr)):
                for track in range(self.numTracks):
                    p = posteriors[i] * (segRatios[i] if segRatios is not None else 1.0)
                    obsStats[track, :, int(arr[i, track])] += p
        return obsStats

    def maximize(self, obsStats, trackList=None):
        """emission.py:243-267 (zeros become -1e6, orphaned state/track rows keep old values)."""
        for track in range(self.numTracks):
            syms = list(self.getTrackSymbols(track))
            for state in range(self.numStates):
                totalSymbol = 0.0
                for symbol in syms:
                    totalSymbol += obsStats[track, state, symbol]
                lastMat = copy.deepcopy(self.logProbs[track][state])
                trackSum = 0
                for symbol in syms:
                    denom = max(self.fudge, totalSymbol)
                    symbolProb = obsStats[track, state, symbol] / denom if denom != 0. else 0.
                    trackSum += symbolProb
                    self.logProbs[track][state][symbol] = myLog(symbolProb, logZeroVal=-1e6)
                if trackSum < EPSILON:
                    self.logProbs[track][state] = lastMat
        self.validate()

    def supervisedTrain(self, trackData, bedIntervals):
        """emission.py:293-330: emission counts of every state from sorted labelled intervals.  The
        reference updates the counts one overlap at a time (fastUpdateCounts); here the overlaps are
        gathered per table, in the same order, and every table is ONE device call."""
        tables = trackData.getTrackTableList()
        assert len(tables) > 0 and len(bedIntervals) > 0
        obsStats = self.initStats()
        per_table = [[] for _ in tables]
        lastHit, lastOverlapEnd = 0, -1
        for interval in bedIntervals:
            hit = False
            for tableIdx in range(lastHit, len(tables)):
                overlap = tables[tableIdx].getOverlapInTableCoords(interval, lastOverlapEnd)
                if overlap is not None:
                    lastHit, hit = tableIdx, True
                    lastOverlapEnd = max(0, overlap[2] - 1)
                    per_table[tableIdx].append(overlap)
                elif hit is True:
                    break
        for table, overlaps in zip(tables, per_table):
            if not overlaps:
                continue
            if canFast(table):
                fastUpdateCountsBatch(overlaps, table, obsStats, self.getSegmentRatios(table))
            else:
                raise TypeError("supervisedTrain needs integer TrackTables")
        self.maximize(obsStats, trackData.getTrackList())
        self.validate()

    def validate(self):
        """emission.py:269-291: every state's distribution over symbol vectors sums to 1."""
        numSymbols = reduce(lambda x, y: max(x, 1) * max(y, 1), self.numSymbolsPerTrack, 1)
        if numSymbols >= 1000 or self.normalizeFac != 1.0:
            return
        allSymbols = [x for x in self.getSymbols()]
        assert len(allSymbols) == numSymbols
        for state in range(self.numStates):
            total = 0.
            for val in allSymbols:
                total += np.exp(self.singleLogProb(state, val))
            if len(allSymbols) > 0:
                assert_array_almost_equal(total, 1.)

    def getSegmentRatios(self, obs):
        """emission.py:473-481: only a segmented TrackTable with an effective length has ratios."""
        if isinstance(obs, TrackTable):
            if obs.getSegmentOffsets() is not None and self.effectiveSegmentLength is not None:
                return obs.getSegmentLengthsAsRatio(self.effectiveSegmentLength)
        return None


class IndependentMultinomialAndGaussianEmissionModel(IndependentMultinomialEmissionModel):
    """emission.py:483-593: gaussian tracks are baked into the same table (mu, sigma estimated from
    the multinomial, table = uniformMix/S + (1 - uniformMix) * normpdf, renormalised)."""

    def __init__(self, numStates, numSymbolsPerTrack, trackList, params=None, zeroAsMissingData=True,
                 fudge=0.0, normalizeFac=0.0, randomize=False, effectiveSegmentLength=None,
                 random_state=None, randRange=(0.1, 0.9)):
        super(IndependentMultinomialAndGaussianEmissionModel, self).__init__(
            numStates, numSymbolsPerTrack, params, zeroAsMissingData, fudge, normalizeFac, randomize,
            effectiveSegmentLength, random_state, randRange)
        self.gaussParams = None
        self.makeGaussian(trackList)

    def makeGaussian(self, trackList):
        self.gaussParams = np.zeros((self.numTracks, self.numStates, 2), dtype=np.float64)
        assert self.numTracks == len(trackList)
        for track in trackList:
            if track.getDist() == "gaussian":
                for state in range(self.numStates):
                    mu, sigma = self.computeMuSigma(track, state)
                    self.gaussParams[track.getNumber(), state, 0] = mu
                    self.gaussParams[track.getNumber(), state, 1] = sigma
                    self.applyGaussian(track, state)

    def computeMuSigma(self, track, state):
        catMap = track.getValueMap()
        trackNo = track.getNumber()
        syms = list(self.getTrackSymbols(trackNo))
        vals = np.asarray([float(catMap.getMapBack(s)) for s in syms])
        probs = np.exp(self.logProbs[trackNo][state][syms])
        mu = 0.
        for v, p in zip(vals, probs):
            mu += v * p
        sigma = 0.
        for v, p in zip(vals, probs):
            sigma += np.square(v - mu) * p
        sigma = np.sqrt(sigma)
        return mu, max(sigma, EPSILON)

    def applyGaussian(self, track, state, logProbs=None):
        catMap = track.getValueMap()
        trackNo = track.getNumber()
        if logProbs is None:
            logProbs = self.logProbs
        uniformProb = 1. / float(self.numSymbolsPerTrack[trackNo])
        uniformProb *= self.uniformMixProb
        for symbol in self.getTrackSymbols(trackNo):
            actualValue = float(catMap.getMapBack(symbol))
            prob = stats.norm.pdf(actualValue, loc=self.gaussParams[trackNo, state, 0],
                                  scale=self.gaussParams[trackNo, state, 1])
            prob = uniformProb + (1. - self.uniformMixProb) * prob
            assert prob > EPSILON
            logProbs[trackNo][state][symbol] = myLog(prob)
        probs = np.exp(logProbs[trackNo][state])
        tot = 0.
        for symbol in self.getTrackSymbols(trackNo):
            tot += probs[symbol]
        assert tot > 0.
        for symbol in self.getTrackSymbols(trackNo):
            logProbs[trackNo][state][symbol] = myLog(probs[symbol] / tot)

    def getGaussianParams(self, trackNo, state):
        return self.gaussParams[trackNo, state]

    def maximize(self, obsStats, trackList):
        super(IndependentMultinomialAndGaussianEmissionModel, self).maximize(obsStats)
        self.makeGaussian(trackList)
