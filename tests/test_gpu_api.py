"""GPU tests of the host-side mirror of teHmm's model API (MultitrackHmm / BaseHMM / emission models):
the reference's own unit tests (tests/hmmTest.py, tests/emissionTest.py) re-expressed against
tehmm_amd, plus golden vectors produced by the real reference for the driver quirks and one EM
iteration."""
import math

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_almost_equal, assert_array_equal

from conftest import load_golden

pytestmark = pytest.mark.gpu

EMISSIONPROB = [[0.1, 0.4, 0.5], [0.6, 0.3, 0.1]]
STARTPROB = [0.6, 0.4]
TRANSMAT = [[0.7, 0.3], [0.4, 0.6]]


def _segmented_table(obs, lens):
    from tehmm_amd.track import IntegerTrackTable
    T, K = obs.shape
    tab = IntegerTrackTable(K, "chrS", 0, int(np.sum(lens)))
    tab.setData(obs)
    tab.setSegmentOffsets(np.concatenate([[0], np.cumsum(lens)[:-1]]))
    return tab


def test_wikipedia_example():
    """hmmTest.py:48-135."""
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    from tehmm_amd.track import IntegerTrackTable
    g = load_golden("wikipedia")
    trackObs = np.asarray([[0], [1], [2]], dtype=np.uint8)
    em = IndependentMultinomialEmissionModel(2, [3], [EMISSIONPROB], zeroAsMissingData=False)
    h = MultitrackHmm(em, startprob=STARTPROB, transmat=TRANSMAT)
    assert_array_equal(h._compute_log_likelihood(trackObs), g["frame"])
    logprob, seq = h.decode(trackObs)
    assert abs(np.exp(logprob) - 0.01344) < 1e-9
    assert_array_equal(seq, [1, 0, 0])
    # three- and four-track variants
    em3 = IndependentMultinomialEmissionModel(2, [3, 1, 1], [EMISSIONPROB, [[1.], [1.]], [[1.], [1.]]],
                                              zeroAsMissingData=False)
    h3 = MultitrackHmm(em3, startprob=STARTPROB, transmat=TRANSMAT)
    obs3 = np.asarray([[0, 0, 0], [1, 0, 0], [2, 0, 0]], dtype=np.uint8)
    assert_array_equal(h3._compute_log_likelihood(obs3), g["frame"])
    logprob, seq = h3.decode(obs3)
    assert abs(np.exp(logprob) - 0.01344) < 1e-9
    assert_array_equal(seq, [1, 0, 0])
    obs4 = np.asarray([[0, 0, 0, 0], [1, 0, 0, 5], [2, 0, 0, 7]], dtype=np.uint8)
    ep4 = [EMISSIONPROB, [[1.], [1.]], [[1.], [1.]], [[.1] * 10, [.1] * 10]]
    em4 = IndependentMultinomialEmissionModel(2, [3, 1, 1, 10], ep4, zeroAsMissingData=False)
    h4 = MultitrackHmm(em4, startprob=STARTPROB, transmat=TRANSMAT)
    logprob, seq = h4.decode(obs4)
    assert abs(np.exp(logprob) - 0.01344 * 1e-3) < 1e-12
    assert logprob == g["vit_logprob4"]
    assert_array_equal(seq, [1, 0, 0])
    # the same through a TrackTable (hmmTest.py:128-135)
    tab = IntegerTrackTable(4, "scaffold_1", 10, 13)
    for row in range(4):
        tab.writeRow(row, [obs4[0][row], obs4[1][row], obs4[2][row]])
    logprob, seq = h4.decode(tab)
    assert logprob == g["vit_logprob4"]
    assert_array_equal(seq, [1, 0, 0])


def test_predict_and_forward_backward_identity():
    """hmmTest.py:138-191: posteriors known answer; forward table vs the reference; the
    backward total-probability identity."""
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    g = load_golden("wikipedia")
    em3 = IndependentMultinomialEmissionModel(2, [3, 1, 1], [EMISSIONPROB, [[1.], [1.]], [[1.], [1.]]],
                                              zeroAsMissingData=False)
    h3 = MultitrackHmm(em3, startprob=STARTPROB, transmat=TRANSMAT)
    obs3 = np.asarray([[0, 0, 0], [1, 0, 0], [2, 0, 0]], dtype=np.uint8)
    assert_array_equal(h3.predict(obs3), [1, 0, 0])
    assert_allclose(h3.predict_proba(obs3), g["post"], rtol=1e-6)
    emProbs = em3.allLogProbs(obs3)
    flp, ftable = h3._do_forward_pass(emProbs)
    assert_allclose(ftable, g["fwd"], rtol=1e-9)
    assert_allclose(flp, g["fwd_logprob"], rtol=1e-9)
    btable = h3._do_backward_pass(emProbs)
    assert_allclose(btable, g["bwd"], rtol=1e-9)
    bneg1 = np.zeros(2)
    for i in range(2):
        for j in range(2):
            bneg1[i] += np.exp(h3._log_startprob[j] + emProbs[0, j] + btable[0, j])
    assert abs(np.log(np.sum(bneg1)) - flp) < 1e-12


def test_emission_model_known_answers():
    """emissionTest.py:61-105."""
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    em = IndependentMultinomialEmissionModel(numStates=2, numSymbolsPerTrack=[2])
    em.initParams([[[0.2, 0.8], [0.5, 0.5]]])
    assert em.singleLogProb(0, [1]) == math.log(0.2)
    assert em.singleLogProb(1, [0]) == 0
    truth = np.array([[math.log(0.2), math.log(0.5)], [math.log(0.2), math.log(0.5)],
                      [math.log(0.8), math.log(0.5)]])
    assert np.array_equal(em.allLogProbs(np.array([[1], [1], [2]], dtype=np.uint8)), truth)
    assert np.array_equal(em.allLogProbs(np.array([[1], [1], [2]], dtype=np.int32)), truth)
    obsStats = em.initStats()
    assert obsStats[0].shape == (2, 3)
    obs = np.array([[0], [0], [1]], dtype=np.uint8)
    posteriors = np.array([[0.01, 0.02], [0.01, 0.02], [0.3, 0.4]])
    em.accumulateStats(obs, obsStats, posteriors)
    assert obsStats[0][0][0] == 0.01 + 0.01
    assert obsStats[0][1][0] == 0.02 + 0.02
    assert obsStats[0][0][1] == 0.3
    assert obsStats[0][1][1] == 0.4


def _golden_hmm(g, eff_len=None, **kw):
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    K, N, S = g["log_probs"].shape
    syms = [int(x) for x in g["symbols"]] if "symbols" in g else [S - 1] * K
    em = IndependentMultinomialEmissionModel(N, syms, effectiveSegmentLength=eff_len)
    em.logProbs = g["log_probs"].copy()
    h = MultitrackHmm(em, **kw)
    h.transmat_ = g["transmat"].copy()
    h.startprob_ = g["startprob"].copy()
    return h, em


def test_driver_asymmetry_quirks():
    """Q11 / Q12 / Q14 on a segmented TrackTable, against what the real reference returned."""
    g = load_golden("driver_asymmetry")
    h, em = _golden_hmm(g, eff_len=int(g["eff_len"]))
    assert_array_equal(h._log_transmat, g["lt"])
    tab = _segmented_table(g["obs"], g["seg_lens"])
    assert_array_equal(em.getSegmentRatios(tab), g["ratios"])
    lp, path = h.decode(tab)
    assert_array_equal(path, g["decode_path"])
    assert lp == g["decode_logprob"]
    lp, path = h.decode(tab, algorithm="map")          # still Viterbi (Q14)
    assert_array_equal(path, g["map_path"])
    assert lp == g["map_logprob"]
    lp, post = h.score_samples(tab)
    assert_allclose(post, g["score_post"], rtol=1e-6, atol=1e-15)
    assert_allclose(lp, g["score_logprob"], rtol=1e-9)
    assert_array_equal(h._compute_log_likelihood(tab), g["frame_with_ratios"])
    # batch API over several tables at once, mixed with an unsegmented one
    from tehmm_amd.track import IntegerTrackTable, TrackData
    plain = IntegerTrackTable(g["obs"].shape[1], "chrP", 0, 200).setData(g["obs"][:200])
    td = TrackData([tab, plain, tab])
    out = h.viterbi(td)
    assert out[0][0] == g["decode_logprob"] and out[2][0] == g["decode_logprob"]
    assert_array_equal(out[0][1], g["decode_path"])
    lp_p, path_p = h.decode(plain)
    assert out[1][0] == lp_p
    assert_array_equal(out[1][1], path_p)
    posts = h.posteriorDistribution(td)
    assert_allclose(posts[0], g["score_post"], rtol=1e-6, atol=1e-15)
    assert posts[1].shape == (200, h.n_components)


@pytest.mark.parametrize("with_ratio", [0, 1])
def test_em_iteration_matches_reference(with_ratio, monkeypatch):
    """One Baum-Welch iteration (E-step statistics and the parameters after the M-step) against
    what the real reference computed on the same three sequences (one of length 1).  Host M-step
    path (the statistics are inspected on the host); the device-resident loop is held to the same
    fixture in tests/test_gpu_r2.py."""
    monkeypatch.setenv("TEHMM_DEVICE_EM", "0")
    g = load_golden("em_iteration_r%d" % with_ratio)
    h, em = _golden_hmm(g, eff_len=(int(g["eff_len"]) if with_ratio else None), n_iter=2, thresh=0.0,
                        fixStart=False, fudge=0.0)
    seqs = [g["obs%d" % i] for i in range(3)]
    tabs = [_segmented_table(s, g["seg_lens%d" % i]) for i, s in enumerate(seqs)] if with_ratio else seqs
    captured = {}
    orig = h._do_mstep

    def spy(stats, params):
        if "start" not in captured:
            captured.update({k: np.copy(v) for k, v in stats.items()})
        return orig(stats, params)
    h._do_mstep = spy
    h.init_params = ""
    h.fit(tabs)
    assert captured["nobs"] == g["stats_nobs"]
    assert_allclose(captured["start"], g["stats_start"], rtol=1e-6)
    assert_allclose(captured["trans"], g["stats_trans"], rtol=1e-6, atol=1e-300)
    assert_allclose(captured["obs"], g["stats_obs"], rtol=1e-6, atol=1e-300)
    assert_allclose(h.transmat_, g["transmat_after"], rtol=1e-6)
    assert_allclose(h.startprob_, g["startprob_after"], rtol=1e-6)
    assert_allclose(h.emissionModel.logProbs, g["log_probs_after"], rtol=1e-6, atol=1e-9)
    assert_allclose(h.last_forward_log_prob, g["last_logprob"], rtol=1e-9)


def test_dpbenchmark_frame_through_hooks():
    """tests/dpBenchmark.py: the makeFrame(5, 1e5) frame through the three DP hooks, with and
    without its random segment ratios (config 1 of BASELINE.json)."""
    import random
    from tehmm_amd.common import myLog
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    g = load_golden("dpbenchmark_s5_n100000")
    S, n = 5, 100000
    frame = np.asarray(myLog(np.arange(S) / float(S)))[None, :] + \
        np.asarray(myLog((np.arange(n) % 9 + 1.0) / 10))[:, None]
    h = MultitrackHmm(emissionModel=IndependentMultinomialEmissionModel(S, [2]))
    lp, path = h._do_viterbi_pass(frame)
    assert_array_equal(path, g["vit_path"].astype(np.int64))
    assert lp == g["vit_logprob"]
    flp, ftab = h._do_forward_pass(frame)
    assert_allclose(flp, g["fwd_logprob"], rtol=1e-9)
    assert_allclose(ftab[-1], g["fwd_last"], rtol=1e-9)
    assert_allclose(h._do_backward_pass(frame)[0], g["bwd_first"], rtol=1e-9)
    random.seed(200)
    segr = np.asarray([random.uniform(0.01, 10.) for _ in range(n)])
    h.emissionModel.getSegmentRatios = lambda x: segr
    lp, path = h._do_viterbi_pass(frame)
    assert_array_equal(path, g["vit_path_r"].astype(np.int64))
    assert lp == g["vit_logprob_r"]
    flp, ftab = h._do_forward_pass(frame)
    assert_allclose(flp, g["fwd_logprob_r"], rtol=1e-9)
    assert_allclose(h._do_backward_pass(frame)[0], g["bwd_first_r"], rtol=1e-9)
