"""Driver base class of the HMM API -- host-side mirror of the reference's ``basehmm.py``
(scikit-learn derived BaseHMM): decode / score_samples / score / predict / fit and the
sufficient-statistics bookkeeping.  All T-sized work happens in the HIP library through the hook
methods (``_do_viterbi_pass`` etc.), which ``MultitrackHmm`` overrides.
"""
import copy
import numbers
import string

import numpy as np

from . import _hmm
from .common import EPSILON, NEGINF, ZEROLOGPROB, F32_EPS, logger, logsumexp, normalize

decoder_algorithms = ("viterbi", "map")


def check_random_state(seed):
    """basehmm.py:96-111."""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (numbers.Integral, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError('%r cannot be used to seed a numpy.random.RandomState instance' % seed)


class BaseHMM(object):
    def __init__(self, n_components=1, startprob=None, transmat=None, startprob_prior=None,
                 transmat_prior=None, algorithm="viterbi", random_state=None, n_iter=10, thresh=1e-2,
                 params=string.ascii_letters, init_params=string.ascii_letters):
        self.n_components = n_components
        self.n_iter = n_iter
        self.thresh = thresh
        self.params = params
        self.init_params = init_params
        self.startprob_ = startprob
        self.startprob_prior = startprob_prior
        self.transmat_ = transmat
        self.transmat_prior = transmat_prior
        self._algorithm = algorithm
        self.random_state = random_state

    # ------------------------------------------------------------------ inference drivers
    def eval(self, X):
        return self.score_samples(X)

    def score_samples(self, obs):
        """basehmm.py:238-273: (logprob, posteriors); np.asarray strips a TrackTable, so segment
        ratios never apply here (quirk Q12); float32 eps is added and rows renormalised."""
        obs = self._as_array(obs)
        framelogprob = self._compute_log_likelihood(obs)
        logprob, fwdlattice = self._do_forward_pass(framelogprob)
        bwdlattice = self._do_backward_pass(framelogprob)
        gamma = fwdlattice + bwdlattice
        posteriors = np.exp(gamma.T - logsumexp(gamma, axis=1)).T
        posteriors += F32_EPS
        posteriors /= np.sum(posteriors, axis=1).reshape((-1, 1))
        return logprob, posteriors

    def score(self, obs):
        obs = self._as_array(obs)
        framelogprob = self._compute_log_likelihood(obs)
        logprob, _ = self._do_forward_pass(framelogprob)
        return logprob

    def _decode_viterbi(self, obs):
        """basehmm.py:301-330: emission on the stripped array (no ratios), Viterbi with obs=table
        (ratios on transitions) -- quirk Q11."""
        framelogprob = self._compute_log_likelihood(self._as_array(obs))
        viterbi_logprob, state_sequence = self._do_viterbi_pass(framelogprob, obs=obs)
        return viterbi_logprob, state_sequence

    def _decode_map(self, obs):
        """basehmm.py:332-359 ("logprob" = sum of the row maxima, quirk Q13)."""
        _, posteriors = self.score_samples(obs)
        state_sequence = np.argmax(posteriors, axis=1)
        map_logprob = np.max(posteriors, axis=1).sum()
        return map_logprob, state_sequence

    def decode(self, obs, algorithm="viterbi"):
        """basehmm.py:361-396: the MODEL's algorithm wins over the argument (quirk Q14)."""
        if self._algorithm in decoder_algorithms:
            algorithm = self._algorithm
        elif algorithm in decoder_algorithms:
            algorithm = algorithm
        decoder = {"viterbi": self._decode_viterbi, "map": self._decode_map}
        logprob, state_sequence = decoder[algorithm](obs)
        return logprob, state_sequence

    def predict(self, obs, algorithm="viterbi"):
        _, state_sequence = self.decode(obs, algorithm)
        return state_sequence

    def predict_proba(self, obs):
        _, posteriors = self.score_samples(obs)
        return posteriors

    def sample(self, n=1, random_state=None):
        """basehmm.py:431-473."""
        if random_state is None:
            random_state = self.random_state
        random_state = check_random_state(random_state)
        startprob_cdf = np.cumsum(self.startprob_)
        transmat_cdf = np.cumsum(self.transmat_, 1)
        rand = random_state.rand()
        currstate = (startprob_cdf > rand).argmax()
        hidden_states = [currstate]
        obs = [self._generate_sample_from_state(currstate, random_state=random_state)]
        for _ in range(n - 1):
            rand = random_state.rand()
            currstate = (transmat_cdf[currstate] > rand).argmax()
            hidden_states.append(currstate)
            obs.append(self._generate_sample_from_state(currstate, random_state=random_state))
        return np.array(obs), np.array(hidden_states, dtype=int)

    # ------------------------------------------------------------------ Baum-Welch
    def fit(self, obs):
        """basehmm.py:475-541.  E-step per sequence: emission, forward, backward, posteriors (no
        eps), statistics; convergence is checked BEFORE the M-step and the last iteration skips
        it (quirk Q18)."""
        if self.algorithm not in decoder_algorithms:
            self._algorithm = "viterbi"
        self._init(obs, self.init_params)
        logprob = []
        for i in range(copy.deepcopy(self.n_iter)):
            stats = self._initialize_sufficient_statistics()
            curr_logprob = self._do_estep(obs, stats)
            logprob.append(curr_logprob)
            msg = "BW Iteration %d: LogProb %f" % (i, curr_logprob)
            if i > 0:
                msg += " (delta %f)" % (logprob[-1] - logprob[-2])
            logger.info(msg)
            if i > 0 and abs(logprob[-1] - logprob[-2]) < self.thresh:
                break
            if i == self.n_iter - 1:
                break
            self._do_mstep(stats, self.params)
        return self

    def _do_estep(self, obs, stats):
        """The per-sequence loop of basehmm.py:507-522."""
        curr_logprob = 0
        for seq in obs:
            framelogprob = self._compute_log_likelihood(seq)
            lpr, fwdlattice = self._do_forward_pass(framelogprob, obs=seq)
            bwdlattice = self._do_backward_pass(framelogprob, obs=seq)
            gamma = fwdlattice + bwdlattice
            posteriors = np.exp(gamma.T - logsumexp(gamma, axis=1)).T
            curr_logprob += lpr
            self._accumulate_sufficient_statistics(stats, seq, framelogprob, posteriors, fwdlattice,
                                                   bwdlattice, self.params)
        return curr_logprob

    # ------------------------------------------------------------------ properties
    def _get_algorithm(self):
        return self._algorithm

    def _set_algorithm(self, algorithm):
        if algorithm not in decoder_algorithms:
            raise ValueError("algorithm must be one of the decoder_algorithms")
        self._algorithm = algorithm

    algorithm = property(_get_algorithm, _set_algorithm)

    def _get_startprob(self):
        return np.exp(self._log_startprob)

    def _set_startprob(self, startprob):
        """basehmm.py:560-577 (zeros get machine eps, unlike MultitrackHmm)."""
        if startprob is None:
            startprob = np.tile(1.0 / self.n_components, self.n_components)
        else:
            startprob = np.asarray(startprob, dtype=np.float64)
        if not np.all(startprob):
            startprob = normalize(startprob)
        if len(startprob) != self.n_components:
            raise ValueError('startprob must have length n_components')
        if not np.allclose(np.sum(startprob), 1.0):
            raise ValueError('startprob must sum to 1.0')
        self._log_startprob = np.log(np.asarray(startprob).copy())

    startprob_ = property(_get_startprob, _set_startprob)

    def _get_transmat(self):
        return np.exp(self._log_transmat)

    def _set_transmat(self, transmat):
        """basehmm.py:585-603."""
        if transmat is None:
            transmat = np.tile(1.0 / self.n_components, (self.n_components, self.n_components))
        transmat = np.asarray(transmat, dtype=np.float64).copy()
        if not np.all(transmat):
            transmat = normalize(transmat, axis=1)
        if transmat.shape != (self.n_components, self.n_components):
            raise ValueError('transmat must have shape (n_components, n_components)')
        if not np.all(np.allclose(np.sum(transmat, axis=1), 1.0)):
            raise ValueError('Rows of transmat must sum to 1.0')
        with np.errstate(divide="ignore"):
            self._log_transmat = np.log(transmat)
        self._log_transmat[np.isnan(self._log_transmat)] = NEGINF

    transmat_ = property(_get_transmat, _set_transmat)

    # ------------------------------------------------------------------ hooks
    @staticmethod
    def _as_array(obs):
        """np.asarray(TrackTable) of the reference, without its O(T) Python row loop (quirk Q16)."""
        from .track import TrackTable
        if isinstance(obs, TrackTable):
            return obs.getNumPyArray()
        return np.asarray(obs)

    def _do_viterbi_pass(self, framelogprob, obs=None):
        n_observations, n_components = framelogprob.shape
        state_sequence, logprob = _hmm._viterbi(n_observations, n_components, self._log_startprob,
                                                self._log_transmat, None,
                                                np.ascontiguousarray(framelogprob))
        return logprob, state_sequence

    def _do_forward_pass(self, framelogprob, obs=None):
        n_observations, n_components = framelogprob.shape
        fwdlattice = np.zeros((n_observations, n_components))
        _hmm._forward(n_observations, n_components, self._log_startprob, self._log_transmat,
                      np.ascontiguousarray(framelogprob), None, fwdlattice)
        fwdlattice[fwdlattice <= ZEROLOGPROB] = NEGINF
        return logsumexp(fwdlattice[-1]), fwdlattice

    def _do_backward_pass(self, framelogprob, obs=None):
        n_observations, n_components = framelogprob.shape
        bwdlattice = np.zeros((n_observations, n_components))
        _hmm._backward(n_observations, n_components, self._log_startprob, self._log_transmat,
                       np.ascontiguousarray(framelogprob), None, bwdlattice)
        bwdlattice[bwdlattice <= ZEROLOGPROB] = NEGINF
        return bwdlattice

    def _compute_log_likelihood(self, obs):
        pass

    def _generate_sample_from_state(self, state, random_state=None):
        pass

    def _init(self, obs, params):
        if 's' in params:
            self.startprob_.fill(1.0 / self.n_components)
        if 't' in params:
            self.transmat_.fill(1.0 / self.n_components)

    def _initialize_sufficient_statistics(self):
        return {'nobs': 0, 'start': np.zeros(self.n_components),
                'trans': np.zeros((self.n_components, self.n_components))}

    def _accumulate_sufficient_statistics(self, stats, seq, framelogprob, posteriors, fwdlattice,
                                          bwdlattice, params):
        stats['nobs'] += 1
        if 's' in params:
            stats['start'] += posteriors[0]
        if 't' in params:
            n_observations, n_components = framelogprob.shape
            if n_observations > 1:
                out = np.zeros((n_components, n_components))
                lnP = logsumexp(fwdlattice[-1])
                _hmm._log_sum_lneta(n_observations, n_components, fwdlattice, self._log_transmat,
                                    bwdlattice, framelogprob, lnP, None, out)
                stats["trans"] += np.exp(out)

    def _do_mstep(self, stats, params):
        """basehmm.py:660-676."""
        if self.startprob_prior is None:
            self.startprob_prior = 1.0
        if self.transmat_prior is None:
            self.transmat_prior = 1.0
        if 's' in params:
            self.startprob_ = normalize(np.maximum(self.startprob_prior - 1.0 + stats['start'], 1e-20))
        if 't' in params:
            self.transmat_ = normalize(np.maximum(self.transmat_prior - 1.0 + stats['trans'], 1e-20),
                                       axis=1)
