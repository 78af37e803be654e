"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host-side helpers
behave like the reference's, the oracle is thread-safe.  No GPU compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tehmm_amd import _lib, build
    build.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "tehmm_hip.h")).read()
    declared = set(re.findall(r"\b(tehmm_[a-z0-9_]+)\s*\(", header))
    declared -= {"tehmm_max_states)"}
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libtehmm_hip.so does not export %s" % name
    assert set(_lib.SIGNATURES) == declared
    assert lib.tehmm_abi_version() >= 1
    assert lib.tehmm_max_states() == 128


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device the array-level entry points must fail with an error code (and the Python
    wrappers raise); nothing silently computes on the CPU."""
    from tehmm_amd import _lib, _hmm
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    frame = np.zeros((4, 2))
    with pytest.raises(_lib.TeHmmHipError):
        _hmm._viterbi(4, 2, np.zeros(2), np.zeros((2, 2)), None, frame)
    with pytest.raises(ValueError):
        _hmm._viterbi(4, 2, np.zeros(2, dtype=np.float32), np.zeros((2, 2)), None, frame)


def test_missing_library_raises():
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['TEHMM_HIP_LIB']='/nonexistent/lib.so';"
            "from tehmm_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.TeHmmHipError as e:\n    print('RAISED'); sys.exit(0)\n"
            "sys.exit(1)") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "RAISED" in out.stdout


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tehmm_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\.*oracle", src, re.M), "%s imports the oracle" % fn
            assert "libtehmm_oracle" not in src and "oracle." not in src, "%s uses the oracle" % fn


def test_mylog_and_logsumexp():
    from tehmm_amd.common import LOGZERO, logsumexp, myLog, normalize
    a = np.asarray(myLog(np.array([[0.5, 0.0], [1.0, 1e-17]])))
    assert a[0, 1] == LOGZERO == -1e100 and a[1, 1] == LOGZERO
    assert a[0, 0] == np.log(0.5) and a[1, 0] == 0.0
    assert myLog(0.0, logZeroVal=-1e6) == -1e6
    x = np.array([[-1.0, -2.0, -3.0], [-10.0, -10.0, -10.0]])
    np.testing.assert_allclose(logsumexp(x, axis=1), np.log(np.exp(x).sum(axis=1)), rtol=1e-15)
    p = normalize(np.array([1.0, 3.0]))
    np.testing.assert_allclose(p, [0.25, 0.75], rtol=1e-15)


def test_track_table_contract():
    from tehmm_amd.track import IntegerTrackTable
    tab = IntegerTrackTable(3, "chr1", 100, 110)
    assert tab.getNumPyArray().dtype == np.uint8 and tab.getNumPyArray().shape == (10, 3)
    tab.writeRow(1, np.arange(10) + 250)            # clamps to 255 like track.py:568-580
    assert tab.getNumPyArray()[:, 1].max() == 255
    assert tab.getSegmentLengthsAsRatio(20) is None
    seg = IntegerTrackTable(2, "chr1", 0, 100).setData(np.ones((4, 2), dtype=np.uint8))
    seg.setSegmentOffsets([0, 10, 30, 90])
    assert len(seg) == 4 and seg.shape == (4, 2)
    np.testing.assert_array_equal(seg.getSegmentLengthsAsRatio(20), np.array([10, 20, 60, 10]) / 20.0)
    assert seg.getSegmentLength(3) == 10 and seg.getSegmentLength(1) == 20


def test_emission_model_host_logic():
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    em = IndependentMultinomialEmissionModel(numStates=2, numSymbolsPerTrack=[2, 3])
    em.initParams([[[0.2, 0.8], [0.5, 0.5]], [[0.1, 0.3, 0.6], [0.7, 0.1, 0.2]]])
    assert em.logProbs.shape == (2, 2, 4)
    assert em.singleLogProb(0, [1, 2]) == np.log(0.2) + np.log(0.3)
    assert em.singleLogProb(1, [2, 3]) == np.log(0.5) + np.log(0.2)
    st = em.initStats()
    assert st.shape == (2, 2, 4)
    # M-step: counts -> probabilities; zeros -> -1e6; symbol 0 untouched (emission.py:243-267)
    st[0, 0, 1:3] = [3.0, 1.0]
    st[0, 1, 1:3] = [0.0, 2.0]
    st[1, :, 1:4] = 1.0
    em.maximize(st)
    np.testing.assert_allclose(np.exp(em.logProbs[0, 0, 1:3]), [0.75, 0.25])
    assert em.logProbs[0, 1, 1] == -1e6 and em.logProbs[0, 1, 2] == 0.0
    np.testing.assert_allclose(np.exp(em.logProbs[1, 0, 1:4]), 1 / 3.0)
    assert em.logProbs[0, 0, 0] == 0.0


def test_gaussian_emission_model_table():
    from tehmm_amd.emission import IndependentMultinomialAndGaussianEmissionModel
    from tehmm_amd.track import Track, TrackList
    tl = TrackList([Track("a", 0), Track("g", 1, dist="gaussian")])
    em = IndependentMultinomialAndGaussianEmissionModel(2, [2, 20], tl)
    p = np.exp(em.logProbs[1, :, 1:21])
    np.testing.assert_allclose(p.sum(axis=1), 1.0, rtol=1e-12)
    mu, sigma = em.getGaussianParams(1, 0)
    assert abs(mu - 9.5) < 1e-9 and sigma > 5


def test_synth_generators():
    from tehmm_amd import synth
    m = synth.make_model(35, seed=0)
    assert m.log_probs.shape == (10, 35, 251) and m.log_transmat.shape == (35, 35)
    np.testing.assert_allclose(np.exp(m.log_transmat).sum(axis=1), 1.0)
    obs = synth.sample_obs(m, 500, seed=1)
    assert obs.dtype == np.uint8 and obs.shape == (500, 10)
    for k, sk in enumerate(m.symbols_per_track):
        assert obs[:, k].min() >= 1 and obs[:, k].max() <= sk
    lens = synth.interval_lengths(10_000_000, 200_000, 2_000_000, seed=3)
    assert lens.sum() == 10_000_000 and lens.min() >= 100_000 and lens.max() <= 2_000_000 + 100_000
    ms = synth.make_model(6, (3, 4), (), seed=7, sparse=0.6)
    assert (ms.log_transmat == -1e100).any()
