"""MultitrackHmm -- host-side mirror of the reference's ``hmm.py`` model API, with the hot path on
the MI355X.

Same class / method names and semantics as the reference (train, viterbi, posteriorDecode,
posteriorDistribution, emissionDistribution, decode, score_samples, fit and the six BaseHMM hooks);
zeros in the transition / start tables become -1e100 via myLog (hmm.py:625-666, quirk Q1).

Device routing
  * the hook methods (_do_viterbi_pass, _do_forward_pass, _do_backward_pass,
    _compute_log_likelihood, _accumulate_sufficient_statistics) call the array-level C-ABI entry
    points, one-to-one with the Cython functions they replace;
  * decode / score_samples and the per-TrackData batch methods (viterbi, posteriorDistribution)
    call the fused entry point tehmm_eval_batch: emission fused into the DP, all tables of a
    TrackData evaluated in ONE launch, results bit-identical (paths) / within 1e-6 (posteriors).
There is no CPU fallback: without the HIP library every call raises.
"""
import copy
import string

import numpy as np

from . import _hmm
from .basehmm import BaseHMM, check_random_state
from .common import EPSILON, F32_EPS, logger, logsumexp, myLog, normalize
from .track import TrackTable


class MultitrackHmm(BaseHMM):
    def __init__(self, emissionModel=None, startprob=None, transmat=None, startprob_prior=None,
                 transmat_prior=None, algorithm="viterbi", random_state=None, n_iter=10, thresh=1e-2,
                 params=string.ascii_letters, init_params=string.ascii_letters, state_name_map=None,
                 fudge=0.0, fixTrans=False, fixEmission=False, fixStart=True, forceUserTrans=None,
                 forceUserEmissions=None, forceUserStart=None, transMatEpsilons=False, maxProb=False,
                 maxProbCut=None):
        n_components = emissionModel.getNumStates() if emissionModel is not None else 1
        # (the reference sets this after BaseHMM.__init__, which breaks zero-containing transmat
        # arguments -- quirk Q20; here it simply works)
        self.transMatEpsilons = transMatEpsilons
        self._dev = None
        BaseHMM.__init__(self, n_components=n_components, startprob=startprob, transmat=transmat,
                         startprob_prior=startprob_prior, transmat_prior=transmat_prior,
                         algorithm=algorithm, random_state=random_state, n_iter=n_iter, thresh=thresh,
                         params=params, init_params=init_params)
        self.init_params = init_params
        self.emissionModel = emissionModel
        self.trackList = None
        self.stateNameMap = state_name_map
        self.fudge = fudge
        self.fixTrans = fixTrans
        self.fixEmission = fixEmission
        self.fixStart = fixStart
        self.current_iteration = None
        self.last_forward_log_prob = None
        self.last_forward_log_prob_it = -1
        if forceUserTrans is not None or forceUserEmissions is not None or forceUserStart is not None:
            raise NotImplementedError("forceUser{Trans,Emissions,Start} text files are outside the "
                                      "hot path this package replaces")
        self.forceUserTrans = self.forceUserEmissions = self.forceUserStart = None
        self.maxProb = maxProb
        self.best_forward_log_prob = None
        self.bestCopy = None
        self.maxProbCut = maxProbCut
        self.numZeroInitEdges = 0
        self.numZeroInitStarts = 0

    # ------------------------------------------------------------------ public API (hmm.py:155-277)
    def train(self, trackData):
        """Unsupervised Baum-Welch from the tables of a TrackData (hmm.py:155-172)."""
        self.bestCopy = None
        self.trackList = trackData.getTrackList()
        self.fit(trackData.getTrackTableList())
        if self.maxProb is True:
            assert self.bestCopy is not None
            self.emissionModel = self.bestCopy.emissionModel
            self.transmat_ = self.bestCopy.transmat_
            self._log_transmat = self.bestCopy._log_transmat
            self.startprob_ = self.bestCopy.startprob_
            self.last_forward_log_prob = self.bestCopy.last_forward_log_prob
            self.last_forward_log_prob_it = self.bestCopy.last_forward_log_prob_it
        self.validate()

    def supervisedTrain(self, trackData, bedIntervals):
        """hmm.py:174-210: transition / start counts from sorted labelled intervals
        (chrom, start, end, state), emissions through emissionModel.supervisedTrain."""
        self.trackList = trackData.getTrackList()
        N = self.emissionModel.getNumStates()
        transitionCount = self.fudge + np.zeros((N, N), dtype=np.float64)
        freqCount = self.fudge + np.zeros((N,), dtype=np.float64)
        prevInterval = None
        for interval in bedIntervals:
            state = int(interval[3])
            assert state < N
            transitionCount[state, state] += interval[2] - interval[1] - 1
            freqCount[state] += interval[2] - interval[1]
            if prevInterval is not None and prevInterval[0] == interval[0]:
                if interval[1] < prevInterval[2]:
                    raise RuntimeError("Overlapping or out of order training intervals detected: "
                                       "%s and %s." % (prevInterval, interval))
                elif interval[1] == prevInterval[2]:
                    transitionCount[prevInterval[3], state] += 1
            prevInterval = interval
        for row in range(len(transitionCount)):
            transitionCount[row] /= np.sum(transitionCount[row])
        self.transmat_ = np.copy(transitionCount)
        self._log_transmat = np.asarray(myLog(transitionCount), dtype=np.float64)
        freqCount /= np.sum(freqCount)
        self.startprob_ = freqCount
        self.emissionModel.supervisedTrain(trackData, bedIntervals)
        self.validate()

    def viterbi(self, trackData, numThreads=1):
        """(logprob, states) per table (hmm.py:221-237) -- all tables in one fused launch."""
        assert numThreads == 1
        tables = trackData.getTrackTableList()
        if self._algorithm == "map":
            out = [self.decode(t) for t in tables]
        else:
            res = self._eval_tables(tables, viterbi=True, posterior=False)
            out = list(zip(res["viterbi_logprob"], res["paths"]))
        return [(p, self._name_states(s)) for p, s in out]

    def posteriorDecode(self, trackData, numThreads=1):
        """hmm.py:239-252: decode(algorithm="map") -- which still runs Viterbi unless the model itself
        was built with algorithm="map" (quirk Q14)."""
        if self._algorithm == "map":
            return [(p, self._name_states(s))
                    for p, s in (self.decode(t, algorithm="map") for t in trackData.getTrackTableList())]
        return self.viterbi(trackData, numThreads)

    def posteriorDistribution(self, trackData):
        """Posterior [T, N] per table (hmm.py:254-263) -- all tables in one fused launch."""
        res = self._eval_tables(trackData.getTrackTableList(), viterbi=False, posterior=True)
        return res["posteriors"]

    def emissionDistribution(self, trackData):
        return [self._compute_log_likelihood(t) for t in trackData.getTrackTableList()]

    def getTrackList(self):
        return self.trackList

    def getStateNameMap(self):
        return self.stateNameMap

    def getEmissionModel(self):
        return self.emissionModel

    def getTransitionProbs(self):
        return self.transmat_

    def getStartProbs(self):
        return self.startprob_

    def getNumFreeParameters(self):
        """Number of free, learnable parameters (hmm.py:490-520; feeds the BIC of teHmmEval.py:216-234)."""
        if self.forceUserTrans is not None or self.forceUserStart is not None or \
                self.forceUserEmissions is not None:
            raise RuntimeError("hmm.getNumFreeParamaters() does not yet support forceUsers{Trans,Start,Emissions}"
                               " functionality.  ie only works for completely unsupervised learining")
        numParams = 0
        numStates = self.emissionModel.getNumStates()
        if self.fixTrans is False:
            numParams += numStates * (numStates - 1) - self.numZeroInitEdges
        if self.fixStart is False:
            numParams += numStates - 1 - self.numZeroInitStarts
        if self.fixEmission is False:
            for track in self.trackList:
                trackNo = track.getNumber()
                if track.getDist() == "gaussian":
                    numTrackParams = 2
                else:
                    numTrackParams = self.emissionModel.getNumSymbolsPerTrack()[trackNo] - 1
                numParams += numStates * numTrackParams
        return numParams

    def validate(self):
        assert len(self.startprob_) == self.emissionModel.getNumStates()
        assert not np.isnan(self.startprob_.any())
        assert not np.isnan(self.transmat_.any())
        assert len(self.transmat_) == self.emissionModel.getNumStates()
        np.testing.assert_array_almost_equal(np.sum(self.startprob_), 1.)
        for i in range(len(self.transmat_)):
            np.testing.assert_array_almost_equal(np.sum(self.transmat_[i]), 1.0)
        self.emissionModel.validate()

    # ------------------------------------------------------------------ fused device paths
    def _device_model(self):
        """HipModel for the current parameters (rebuilt when any table changed)."""
        from .engine import HipModel
        em = self.emissionModel
        key = (self._log_transmat.tobytes(), self._log_startprob.tobytes(), em.logProbs.tobytes(),
               float(em.normalizeFac))
        if self._dev is None or self._dev[0] != key:
            self._dev = (key, HipModel(self._log_transmat, self._log_startprob, em.logProbs,
                                       normalize=em.normalizeFac,
                                       symbols_per_track=em.getNumSymbolsPerTrack()))
        return self._dev[1]

    def _can_fuse(self, tables):
        from . import _lib
        if self.n_components > _lib.load().tehmm_max_states():
            return False
        for t in tables:
            a = t.getNumPyArray() if isinstance(t, TrackTable) else t
            if not (isinstance(a, np.ndarray) and a.ndim == 2):
                return False
            if a.dtype == np.uint8:
                continue
            # uint16 / int32 tables (track.py:555: the type follows the largest symbol) whose symbols fit a byte
            if a.dtype not in (np.uint16, np.int32) or (a.size and (a.max() > 255 or a.min() < 0)):
                return False
        return True

    def _eval_tables(self, tables, viterbi, posterior):
        """decode and/or score_samples over a list of tables.  Ratio semantics exactly as the
        reference drivers: emission never sees ratios; Viterbi transitions do (Q11); posteriors
        never (Q12)."""
        if len(tables) == 0:
            return {"viterbi_logprob": np.zeros(0), "paths": [], "forward_logprob": np.zeros(0),
                    "posteriors": []}
        if not self._can_fuse(tables):
            out = {"viterbi_logprob": [], "paths": [], "forward_logprob": [], "posteriors": []}
            for t in tables:
                if viterbi:
                    lp, s = self._decode_viterbi(t)
                    out["viterbi_logprob"].append(lp)
                    out["paths"].append(s)
                if posterior:
                    lp, p = BaseHMM.score_samples(self, t)
                    out["forward_logprob"].append(lp)
                    out["posteriors"].append(p)
            return out
        from .engine import HipBatch
        arrays = [t.getNumPyArray() if isinstance(t, TrackTable) else np.ascontiguousarray(t)
                  for t in tables]
        lens = np.asarray([a.shape[0] for a in arrays], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ratios = [self.emissionModel.getSegmentRatios(t) for t in tables]
        use_ratios = viterbi and any(r is not None for r in ratios)
        rcat = None
        if use_ratios:
            # a table without ratios behaves like ratio 1.0 on transitions only if the reference
            # would skip every ratio branch; keep exactness by evaluating such tables separately
            if any(r is None for r in ratios):
                res_a = self._eval_tables([t for t, r in zip(tables, ratios) if r is not None],
                                          viterbi, posterior)
                res_b = self._eval_tables([t for t, r in zip(tables, ratios) if r is None],
                                          viterbi, posterior)
                return _merge_results(res_a, res_b, [r is not None for r in ratios])
            rcat = np.concatenate(ratios)
        obs = np.concatenate(arrays, axis=0) if len(arrays) > 1 else arrays[0]
        hm = self._device_model()
        hb = HipBatch(obs, offs, rcat)
        res = hm.eval(hb, viterbi=viterbi, posterior=posterior, use_ratios=use_ratios)
        out = dict(res)
        if viterbi:
            p = hb.paths()
            out["paths"] = [p[offs[i]:offs[i + 1]] for i in range(len(lens))]
        if posterior:
            q = hb.posteriors()
            out["posteriors"] = [q[offs[i]:offs[i + 1]] for i in range(len(lens))]
        hb.close()
        return out

    def decode(self, obs, algorithm="viterbi"):
        """BaseHMM.decode (basehmm.py:361-396), fused on the device for the Viterbi case."""
        if self._algorithm in ("viterbi", "map"):
            algorithm = self._algorithm
        if algorithm == "viterbi" and self._can_fuse([obs]):
            res = self._eval_tables([obs], viterbi=True, posterior=False)
            return res["viterbi_logprob"][0], res["paths"][0]
        return BaseHMM.decode(self, obs, algorithm)

    def score_samples(self, obs):
        """BaseHMM.score_samples (basehmm.py:238-273), fused on the device."""
        if self._can_fuse([obs]):
            res = self._eval_tables([obs], viterbi=False, posterior=True)
            lp = res["forward_logprob"][0]
            self._note_forward_logprob(lp)
            return lp, res["posteriors"][0]
        return BaseHMM.score_samples(self, obs)

    def _name_states(self, states):
        if self.stateNameMap is None:
            return states
        names = np.asarray([self.stateNameMap.getMapBack(i) for i in range(self.n_components)],
                           dtype=object)
        return list(names[np.asarray(states)])

    # ------------------------------------------------------------------ BaseHMM overrides (hmm.py:524-729)
    def _compute_log_likelihood(self, obs):
        return self.emissionModel.allLogProbs(obs)

    def _generate_sample_from_state(self, state, random_state=None):
        return None

    def _init(self, obs, params='ste'):
        if self.fixTrans is True:
            self.params = self.params.replace("t", "")
        if self.fixEmission is True:
            self.params = self.params.replace("e", "")
        if self.fixStart is True:
            self.params = self.params.replace("s", "")
        super(MultitrackHmm, self)._init(obs, params=params)
        self.random_state = check_random_state(self.random_state)

    def _initialize_sufficient_statistics(self):
        stats = super(MultitrackHmm, self)._initialize_sufficient_statistics()
        stats['obs'] = self.emissionModel.initStats()
        return stats

    def _accumulate_sufficient_statistics(self, stats, obs, framelogprob, posteriors, fwdlattice,
                                          bwdlattice, params):
        """hmm.py:545-574."""
        stats['nobs'] += 1
        if 's' in params:
            stats['start'] += posteriors[0]
        if 't' in params:
            n_observations, n_components = framelogprob.shape
            if n_observations > 1:
                logsum_lneta = np.zeros((n_components, n_components))
                lnP = logsumexp(fwdlattice[-1])
                _hmm._log_sum_lneta(n_observations, n_components, fwdlattice, self._log_transmat,
                                    bwdlattice, framelogprob, lnP,
                                    self.emissionModel.getSegmentRatios(obs), logsum_lneta)
                stats["trans"] += np.exp(logsum_lneta)
        if 'e' in params:
            self.emissionModel.accumulateStats(obs, stats['obs'], posteriors)

    def _do_estep(self, obs, stats):
        """E-step over all sequences.  When every sequence is a uint8 table the fused device entry
        point (tehmm_estep_batch) does the whole loop of basehmm.py:507-522 in one call."""
        if self._can_fuse(obs) and self.n_components <= 128:
            from ._lib import TeHmmHipError
            try:
                return self._fused_estep(obs, stats)
            except TeHmmHipError as e:
                # 64..128 states run on the item-parallel passes only (tehmm_wide_estep.hip.h): a batch they do not
                # apply to (impossible emission rows, links that never verify, fewer than 1024 rows) takes the
                # reference's per-sequence loop over the array-level entry points
                if e.code != -3 or self.n_components < 64:
                    raise
        return BaseHMM._do_estep(self, obs, stats)

    def _fused_estep(self, tables, stats):
        from .engine import HipBatch
        arrays = [t.getNumPyArray() if isinstance(t, TrackTable) else np.ascontiguousarray(t)
                  for t in tables]
        ratios = [self.emissionModel.getSegmentRatios(t) for t in tables]
        groups = {True: [i for i, r in enumerate(ratios) if r is not None],
                  False: [i for i, r in enumerate(ratios) if r is None]}
        hm = self._device_model()
        N = self.n_components
        start = np.zeros(N)
        trans = np.zeros((N, N))
        obs_stats = np.zeros_like(stats['obs'])
        total_lp = 0.0
        seq_lp = np.zeros(len(tables))
        for has_r, idx in groups.items():
            if not idx:
                continue
            lens = np.asarray([arrays[i].shape[0] for i in idx], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            obs = np.concatenate([arrays[i] for i in idx], axis=0)
            rcat = np.concatenate([ratios[i] for i in idx]) if has_r else None
            hb = HipBatch(obs, offs, rcat)
            total_lp += hm.estep(hb, has_r, start, trans, obs_stats)
            seq_lp[idx] = hb.interval_logprobs()
            hb.close()
        stats['nobs'] += len(tables)
        if 's' in self.params:
            stats['start'] += start
        if 't' in self.params:
            stats['trans'] += trans
        if 'e' in self.params:
            stats['obs'] += obs_stats
        for lp in seq_lp:                    # the reference books every sequence's forward pass in turn
            self._note_forward_logprob(float(lp))
        return total_lp

    def _do_mstep(self, stats, params):
        """hmm.py:576-616."""
        self.validate()
        if self.startprob_prior is None:
            self.startprob_prior = 1.0
        if self.transmat_prior is None:
            self.transmat_prior = 1.0
        if 's' in params:
            self.startprob_ = normalize(np.maximum(self.startprob_prior - 1.0 + stats['start'], 1e-20))
        if 't' in params:
            lastMat = copy.deepcopy(self.transmat_)
            transmat_ = self.transmat_prior - 1.0 + stats['trans']
            for row in range(len(transmat_)):
                rowSum = np.sum(transmat_[row])
                if rowSum < EPSILON:
                    transmat_[row] = lastMat[row]          # orphaned state keeps its old row
                else:
                    transmat_[row] = transmat_[row] / rowSum
            self.transmat_ = transmat_
        if 'e' in params:
            self.emissionModel.maximize(stats['obs'], self.trackList)
        self.current_iteration += 1
        self.validate()

    def fit(self, obs, shard_lengths=None, **kwargs):
        """basehmm.py:475-541.  shard_lengths (multi-GPU only): the row counts of ALL training tables in global order;
        `obs` may then hold None for the tables this rank does not own (dist.lpt_shard(shard_lengths, world)[rank]),
        so that a rank loads only its own shard of an hg19-sized training set."""
        self.current_iteration = 1
        if shard_lengths is not None:
            if len(shard_lengths) != len(obs):
                raise ValueError("fit: shard_lengths must name every table of the global list")
            return self._fit_device(obs, lengths=[int(x) for x in shard_lengths])
        if self._can_fit_on_device(obs):
            return self._fit_device(obs)
        return BaseHMM.fit(self, obs, **kwargs)

    # ------------------------------------------------------------------ device-resident Baum-Welch
    def _can_fit_on_device(self, obs):
        import os
        from .emission import (IndependentMultinomialAndGaussianEmissionModel,
                               IndependentMultinomialEmissionModel)
        em = self.emissionModel
        if os.environ.get("TEHMM_DEVICE_EM", "1") == "0" or self.transMatEpsilons:
            return False
        if type(em) not in (IndependentMultinomialEmissionModel, IndependentMultinomialAndGaussianEmissionModel):
            return False
        if type(em) is IndependentMultinomialAndGaussianEmissionModel and self.trackList is None:
            return False
        # (decided on the GLOBAL table list, which every rank passes alike: no rank may leave the others alone
        #  in the per-iteration all-reduce)
        if self.n_components >= 64 and sum(len(t) for t in obs if t is not None) < 1024:
            return False                               # (the item-parallel passes want 1024 rows)
        return em.zeroAsMissingData is True and self.n_components <= 128 and len(obs) > 0 and self._can_fuse(obs)

    def _fit_device(self, tables, lengths=None):
        """BaseHMM.fit (basehmm.py:475-541) with the observations, the sufficient statistics and the
        parameters resident on the device: per iteration one fused E-step per batch (statistics ADDED
        into one flat device buffer), the convergence test on the returned log-likelihood, and
        tehmm_model_mstep.  The host model is refreshed at the end (every iteration with maxProb, whose
        bookkeeping deep-copies the model).

        With torch.distributed initialised every rank passes the SAME table list: the tables are
        LPT-sharded here (dist.lpt_shard), each rank keeps only its shard on its GPU, and the ranks meet in
        exactly one all-reduce of the statistics buffer per iteration plus one all-gather of the
        per-sequence log-likelihoods, so that the convergence test, the --maxProb bookkeeping (hmm.py:690-711)
        and the M-step see global values and take identical decisions everywhere.  A rank whose shard is
        empty still takes part, with zero statistics."""
        from . import dist as tdist
        from ._lib import TeHmmHipError
        from .engine import DeviceStats, HipBatch
        if self.algorithm not in ("viterbi", "map"):
            self._algorithm = "viterbi"
        world, rank = tdist.world_rank()
        n_seq = len(tables)
        lengths_given = lengths is not None
        fallback = False
        if lengths is None:
            lengths = [len(t) for t in tables]
        # every rank must hold the same GLOBAL list (or its lengths): a caller that passes per-rank shards would train
        # on 1 / world of each shard with colliding sequence indices -- refused on every rank alike
        tdist.check_same_lengths(lengths, "training table list")
        mine = list(range(n_seq))
        if world > 1:
            mine = [int(i) for i in tdist.lpt_shard(lengths, world)[rank]]
        if any(tables[i] is None for i in mine):
            raise ValueError("fit: a table of this rank's shard is None (shard = dist.lpt_shard(shard_lengths, world)[rank])")
        self._init([tables[i] for i in mine] if any(t is None for t in tables) else tables, self.init_params)
        arrays = {i: (tables[i].getNumPyArray() if isinstance(tables[i], TrackTable)
                      else np.ascontiguousarray(tables[i])) for i in mine}
        ratios = {i: self.emissionModel.getSegmentRatios(tables[i]) for i in mine}
        hm = self._device_model()
        batches = []
        for has_r in (True, False):
            idx = [i for i in mine if (ratios[i] is not None) == has_r]
            if not idx:
                continue
            lens = np.asarray([arrays[i].shape[0] for i in idx], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            obs = np.concatenate([arrays[i] for i in idx], axis=0)
            rcat = np.concatenate([ratios[i] for i in idx]) if has_r else None
            batches.append((has_r, idx, HipBatch(obs, offs, rcat)))
        stats = DeviceStats(hm)
        gauss = self._gaussian_spec()
        logprob = []
        try:
            for i in range(copy.deepcopy(self.n_iter)):
                stats.zero()
                loc_idx, loc_lp = [], []
                for has_r, idx, hb in batches:
                    try:
                        hm.estep_device(hb, has_r, stats)
                    except TeHmmHipError as e:
                        # 64..128 states: a batch the item-parallel passes do not take (a row no state can emit, fewer
                        # than 1024 rows, links that never verify).  Before anything has been learned -- and with no
                        # other rank waiting in a collective -- the reference's own loop over the array-level entry
                        # points takes the whole fit
                        if e.code == -3 and i == 0 and world == 1 and lengths_given is False:
                            fallback = True
                            break
                        raise
                    loc_idx.extend(idx)
                    loc_lp.extend(hb.interval_logprobs())
                if fallback:
                    break
                seq_lp = tdist.gather_interval_scalars(loc_idx, loc_lp, n_seq)
                if self.maxProb is True:
                    self._pull_params(hm)
                for lp in seq_lp:                      # global values, global sequence order: same on every rank
                    self._note_forward_logprob(float(lp))
                tdist.allreduce_device_stats(stats)
                curr_logprob = stats.head()[0]
                logprob.append(curr_logprob)
                logger.info("BW Iteration %d: LogProb %f" % (i, curr_logprob))
                if i > 0 and abs(logprob[-1] - logprob[-2]) < self.thresh:
                    break
                if i == self.n_iter - 1:
                    break
                do_e = "e" in self.params
                gp = hm.mstep(stats, "s" in self.params, "t" in self.params, do_e,
                              1.0 if self.startprob_prior is None else self.startprob_prior,
                              1.0 if self.transmat_prior is None else self.transmat_prior,
                              self.emissionModel.fudge, gauss)
                if gp is not None and do_e:
                    for g, k in enumerate(gauss[0]):
                        self.emissionModel.gaussParams[k] = gp[g]
                self.current_iteration += 1
        finally:
            for _, _, hb in batches:
                hb.close()
            self._pull_params(hm)
            stats.close()
        if fallback:
            keep = self.init_params
            self.init_params = ""                      # (_init has run; the parameters are the ones it left)
            self.current_iteration = 1
            try:
                return BaseHMM.fit(self, tables)
            finally:
                self.init_params = keep
        self.validate()
        return self

    def _gaussian_spec(self):
        """(track indices, values [n][S], uniform mix) of the gaussian tracks, or None."""
        em = self.emissionModel
        if not hasattr(em, "gaussParams") or self.trackList is None:
            return None
        tracks = [t for t in self.trackList if t.getDist() == "gaussian"]
        if not tracks:
            return None
        S = em.logProbs.shape[2]
        vals = np.zeros((len(tracks), S), dtype=np.float64)
        for g, t in enumerate(tracks):
            for sym in em.getTrackSymbols(t.getNumber()):
                vals[g, sym] = float(t.getValueMap().getMapBack(sym))
        return [t.getNumber() for t in tracks], vals, em.uniformMixProb

    def _pull_params(self, hm):
        """Host copy of the device-resident parameters (and the cache key that goes with them)."""
        em = self.emissionModel
        lt, pi, lp = hm.get_params(em.logProbs)
        self._log_transmat, self._log_startprob, em.logProbs = lt, pi, lp
        self._dev = ((lt.tobytes(), pi.tobytes(), lp.tobytes(), float(em.normalizeFac)), hm)

    # transmat / startprob keep exact zeros (-> -1e100), hmm.py:622-666
    def _get_transmat(self):
        return np.exp(self._log_transmat)

    def _set_transmat(self, transmat):
        if transmat is None:
            transmat = np.tile(1.0 / self.n_components, (self.n_components, self.n_components))
        transmat = np.asarray(transmat, dtype=np.float64)
        if not np.all(transmat) and self.transMatEpsilons is True:
            transmat = normalize(transmat.copy(), axis=1)
        if transmat.shape != (self.n_components, self.n_components):
            raise ValueError('transmat must have shape (n_components, n_components)')
        if not np.all(np.allclose(np.sum(transmat, axis=1), 1.0)):
            raise ValueError('Rows of transmat must sum to 1.0')
        self._log_transmat = np.asarray(myLog(transmat.copy()), dtype=np.float64)

    transmat_ = property(_get_transmat, _set_transmat)

    def _get_startprob(self):
        return np.exp(self._log_startprob)

    def _set_startprob(self, startprob):
        if startprob is None:
            startprob = np.tile(1.0 / self.n_components, self.n_components)
        else:
            startprob = np.asarray(startprob, dtype=np.float64)
        if len(startprob) != self.n_components:
            raise ValueError('startprob must have length n_components')
        if not np.allclose(np.sum(startprob), 1.0):
            raise ValueError('startprob must sum to 1.0')
        self._log_startprob = np.asarray(myLog(np.asarray(startprob).copy()), dtype=np.float64)

    startprob_ = property(_get_startprob, _set_startprob)

    def _do_viterbi_pass(self, framelogprob, obs=None):
        n_observations, n_components = framelogprob.shape
        state_sequence, logprob = _hmm._viterbi(
            n_observations, n_components, self._log_startprob, self._log_transmat,
            self.emissionModel.getSegmentRatios(obs), np.ascontiguousarray(framelogprob))
        return logprob, state_sequence

    def _note_forward_logprob(self, lp):
        """EM best-iteration bookkeeping of hmm.py:690-711 (quirk Q17), called once per sequence in
        sequence order.  The reference is Python 2, where `float > None` is True and `None > x` is
        False: best_forward_log_prob starts as None and is seeded at the second iteration."""
        if self.last_forward_log_prob_it != self.current_iteration:
            if self.maxProb is True and (self.current_iteration == 1 or
                                         _py2_gt(self.last_forward_log_prob, self.best_forward_log_prob)):
                self.best_forward_log_prob = self.last_forward_log_prob
                dev, self._dev = self._dev, None
                self.bestCopy = copy.deepcopy(self)
                self._dev = dev
            self.last_forward_log_prob = lp
            self.last_forward_log_prob_it = self.current_iteration
            if (self.maxProb is True and self.bestCopy is not None and self.maxProbCut is not None
                    and self.current_iteration - self.bestCopy.current_iteration > self.maxProbCut):
                logger.info("Stopping due to --maxProbCut %d" % self.maxProbCut)
                self.n_iter = self.current_iteration
        else:
            self.last_forward_log_prob += lp
            if self.maxProb is True and self.current_iteration > 1 and \
                    _py2_gt(self.last_forward_log_prob, self.best_forward_log_prob):
                self.best_forward_log_prob = self.last_forward_log_prob
                dev, self._dev = self._dev, None
                self.bestCopy = copy.deepcopy(self)
                self._dev = dev

    def _do_forward_pass(self, framelogprob, obs=None):
        n_observations, n_components = framelogprob.shape
        fwdlattice = np.zeros((n_observations, n_components))
        _hmm._forward(n_observations, n_components, self._log_startprob, self._log_transmat,
                      np.ascontiguousarray(framelogprob), self.emissionModel.getSegmentRatios(obs),
                      fwdlattice)
        lp = logsumexp(fwdlattice[-1])
        self._note_forward_logprob(lp)
        return lp, fwdlattice

    def _do_backward_pass(self, framelogprob, obs=None):
        n_observations, n_components = framelogprob.shape
        bwdlattice = np.zeros((n_observations, n_components))
        _hmm._backward(n_observations, n_components, self._log_startprob, self._log_transmat,
                       np.ascontiguousarray(framelogprob), self.emissionModel.getSegmentRatios(obs),
                       bwdlattice)
        return bwdlattice

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_dev"] = None            # device handles are not picklable / copyable
        return d


def _py2_gt(a, b):
    """`a > b` with Python 2's ordering of None (None sorts below every number)."""
    if a is None:
        return False
    if b is None:
        return True
    return a > b


def _merge_results(res_a, res_b, mask):
    out = {}
    for k in set(res_a) | set(res_b):
        a, b = res_a.get(k), res_b.get(k)
        if a is None or b is None:
            out[k] = None
            continue
        ia = ib = 0
        merged = []
        for m in mask:
            if m:
                merged.append(a[ia])
                ia += 1
            else:
                merged.append(b[ib])
                ib += 1
        out[k] = np.asarray(merged) if k.endswith("logprob") else merged
    return out
