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


def test_fused_path_takes_the_three_observation_types_of_the_reference():
    """IntegerTrackTable data is uint8, uint16 or int32 (track.py:555, the type follows the largest symbol): the fused
    path takes all three while the symbols fit a byte (tehmm_batch_create / _u16 / _i32), anything else stays on the
    array-level entry points.  The narrowing entry points refuse a symbol beyond 255 before any device call."""
    from tehmm_amd import _lib
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    h = MultitrackHmm(IndependentMultinomialEmissionModel(4, [3, 5]))
    a = np.array([[1, 2], [3, 5], [0, 1]])
    assert h._can_fuse([a.astype(np.uint8)]) and h._can_fuse([a.astype(np.uint16)]) and h._can_fuse([a.astype(np.int32)])
    assert not h._can_fuse([a.astype(np.int64)]) and not h._can_fuse([a.astype(np.float64)])
    big = a.astype(np.uint16)
    big[1, 0] = 256
    neg = a.astype(np.int32)
    neg[0, 0] = -1
    assert not h._can_fuse([big]) and not h._can_fuse([neg]) and not h._can_fuse([a.astype(np.uint8), big])
    lib = _lib.load()
    assert lib.tehmm_abi_version() >= 3
    offs = np.asarray([0, 3], dtype=np.int64)
    out = ctypes.c_void_p()
    for fn, arr in ((lib.tehmm_batch_create_u16, big), (lib.tehmm_batch_create_i32, neg)):
        rc = fn(1, offs.ctypes.data_as(_lib.i64p), 2, arr.ctypes.data_as(ctypes.c_void_p), None, ctypes.byref(out))
        assert rc == -3 and b"outside 0..255" in lib.tehmm_last_error()         # TEHMM_ERR_UNSUPPORTED, no device touched


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


def test_maxprob_bookkeeping_python2_ordering():
    """hmm.py:690-711 under Python 2: `float > None` is True, so best_forward_log_prob is seeded at the
    second iteration and the best copy keeps following improvements (it used to stay None)."""
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm, _py2_gt
    assert _py2_gt(-5.0, None) and not _py2_gt(None, -5.0) and not _py2_gt(None, None)
    h = MultitrackHmm(IndependentMultinomialEmissionModel(2, [2]), maxProb=True)
    # three iterations, two sequences each; iteration totals -100, -90, -95
    per_iter = [(-60.0, -40.0), (-50.0, -40.0), (-55.0, -40.0)]
    marks = []
    for it, lps in enumerate(per_iter, start=1):
        h.current_iteration = it
        for lp in lps:
            h._note_forward_logprob(lp)
        marks.append((h.best_forward_log_prob, h.bestCopy.current_iteration if h.bestCopy else None))
    # iteration 1: best = None (the copy exists); iteration 2: first sequence seeds best with iteration 1's
    # total (-100), the second sequence's partial-sum test (-90 > -100) moves it to -90 (the reference's
    # "very ugly" repeat); iteration 3: -90 (previous total) is not > -90 and -95 is not either
    assert marks[0] == (None, 1)
    assert marks[1] == (-90.0, 2)
    assert marks[2] == (-90.0, 2)
    assert h.last_forward_log_prob == -95.0 and h.last_forward_log_prob_it == 3


def test_native_bed_writer(tmp_path):
    """tehmm_write_bed is host code (no GPU): the lines of teHmmEval.statesToBed, Python-2 float text."""
    from tehmm_amd import output
    starts = np.asarray([10, 20, 35], dtype=np.int64)
    ends = np.asarray([20, 35, 36], dtype=np.int64)
    p = tmp_path / "a.bed"
    output._write(str(p), False, "chr2", starts, ends, states=[2, 0, 1], names=["LTR", "Outside", "TSD"])
    assert p.read_text() == "chr2\t10\t20\tTSD\nchr2\t20\t35\tLTR\nchr2\t35\t36\tOutside\n"
    output._write(str(p), True, "chr2", starts[:1], ends[:1], states=[7])
    assert p.read_text().endswith("chr2\t10\t20\t7\n")
    q = tmp_path / "b.bed"
    vals = [1.0, 0.651120311022880, 1.5e-07]
    output._write(str(q), False, "c", starts, ends, values=vals)
    cols = [ln.split("\t")[3] for ln in q.read_text().strip().split("\n")]
    assert cols == ["1.0", "0.651120311023", "1.5e-07"]      # str(float) of Python 2: '%.12g' (+ '.0')


def test_category_map_and_overlap():
    from tehmm_amd.track import CategoryMap, IntegerTrackTable
    m = CategoryMap(reserved=2)
    assert [m.getMap(x, update=True) for x in ("b", "a", "b", "c")] == [2, 3, 2, 4]
    assert m.getMap("zzz") == m.getMissingVal() == 1 and len(m) == 4
    m.sort()
    assert [m.getMap(x) for x in ("a", "b", "c")] == [2, 3, 4] and m.getMapBack(3) == "b"
    g = CategoryMap(reserved=1, defaultVal="0", scale=0.1)
    assert g.getMap(57.0, update=True) == 2 and g.getMap(51.0) == 2 and g.getMapBack(2) == 50.0
    assert g.getMapBack(99) == 0.0                       # unknown symbol -> the default value
    tab = IntegerTrackTable(1, "c", 100, 200)
    assert tab.getOverlapInTableCoords(("c", 50, 120, 3)) == ["c", 0, 20, 3]
    assert tab.getOverlapInTableCoords(("c", 200, 220, 3)) is None
    assert tab.getOverlapInTableCoords(("d", 100, 120, 3)) is None
    seg = IntegerTrackTable(1, "c", 100, 200).setData(np.ones((4, 1), dtype=np.uint8))
    seg.setSegmentOffsets([0, 10, 30, 90])
    assert seg.getOverlapInTableCoords(("c", 105, 131, 1)) == ["c", 0, 3, 1]
    assert seg.getOverlapInTableCoords(("c", 110, 130, 1)) == ["c", 1, 2, 1]
    assert seg.getOverlapInTableCoords(("c", 195, 400, 1)) == ["c", 3, 4, 1]


def test_oracle_under_address_sanitizer():
    """The CPU oracle rebuilt with -fsanitize=address (oracle/Makefile: `make asan`) runs a ragged batch
    clean (GPU sanitizers are not available on the pool; the checker at least is memory-clean)."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-s", "-C", odir, "asan"])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    code = r'''
import ctypes, numpy as np, sys
sys.path.insert(0, %r)
from tehmm_amd import synth
lib = ctypes.CDLL(%r)
m = synth.make_model(7, (3, 4), (), seed=1)
lens = [1, 2, 130, 65]
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = synth.sample_obs(m, int(offs[-1]), seed=2, missing=0.05)
K, N, S = m.log_probs.shape
paths = np.zeros(int(offs[-1]), dtype=np.int64); vlp = np.zeros(4); flp = np.zeros(4)
post = np.zeros((int(offs[-1]), N))
P = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
rc = lib.oracle_eval_batch(4, P(offs, ctypes.c_int64), K, N, S, P(obs, ctypes.c_uint8),
    P(np.ascontiguousarray(m.log_probs), ctypes.c_double), ctypes.c_double(1.0),
    P(m.log_startprob, ctypes.c_double), P(np.ascontiguousarray(m.log_transmat), ctypes.c_double), None,
    P(paths, ctypes.c_int64), P(vlp, ctypes.c_double), P(flp, ctypes.c_double), P(post, ctypes.c_double), 2)
assert rc == 0 and np.isfinite(vlp).all() and abs(post.sum(axis=1) - 1).max() < 1e-9
print("ASAN_OK")
''' % (ROOT, os.path.join(odir, "libtehmm_oracle_asan.so"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "ASAN_OK" in out.stdout, out.stdout + out.stderr
