"""GPU parity tests of round 2 (run with -m gpu): the rows SURVEY 8(f) marks "next" and the full-size
configurations, through the C ABI, against the reference's golden vectors and the CPU oracle.

Bars: integer / byte work (counts without ratios, modes, mask compaction, BED coordinates, Viterbi
paths) bit-exact; floating point within 1e-6 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-6


@pytest.fixture(scope="module", autouse=True)
def hip():
    from tehmm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1, "no HIP device visible"
    return _lib


def _ratios(g):
    return g["ratios"] if len(g["ratios"]) else None


def _table(obs, chrom, start, seg_lens=None, dtype=np.uint8):
    from tehmm_amd.track import IntegerTrackTable
    total = int(np.sum(seg_lens)) if seg_lens is not None and len(seg_lens) else obs.shape[0]
    tab = IntegerTrackTable(obs.shape[1], chrom, start, start + total, dtype=dtype)
    tab.data = np.ascontiguousarray(obs, dtype=dtype)
    if seg_lens is not None and len(seg_lens):
        tab.setSegmentOffsets(np.concatenate([[0], np.cumsum(seg_lens)[:-1]]))
    tab.shape = tab.data.shape
    return tab


# ------------------------------------------------------------------ 8(f) rank 3 + a7 dtype variants
@pytest.mark.parametrize("tag,dtype", [("u8", np.uint8), ("u16", np.uint16), ("i32", np.int32)])
@pytest.mark.parametrize("r", [0, 1])
def test_update_counts_and_accumulate(tag, dtype, r):
    """fastUpdateCounts / fastAccumulateStats for the three observation dtypes against the reference."""
    from tehmm_amd import _emission
    g = load_golden("counts_%s_r%d" % (tag, r))
    obs, N = g["obs"], int(g["n_states"])
    assert obs.dtype == dtype
    K, S = obs.shape[1], g["stats"].shape[2]
    seg = None
    if r:
        seg = np.round(g["ratios"] * 20.0).astype(np.int64)
    tab = _table(obs, "chr1", 1000, seg, dtype)
    ivs = [("chr1", int(a), int(b), int(s)) for a, b, s in zip(g["iv_start"], g["iv_end"], g["iv_state"])]
    stats = float(g["stats_init"]) + np.zeros((K, N, S))
    for iv in ivs[:3]:                                    # the reference's one-interval-per-call form
        _emission.fastUpdateCounts(iv, tab, stats, _ratios(g))
    _emission.fastUpdateCountsBatch(ivs[3:], tab, stats, _ratios(g))     # the batched form
    assert_array_equal(stats, g["stats"])                 # same accumulation order: bit-identical
    acc = np.zeros((K, N, S))
    _emission.fastAccumulateStats(obs, acc, g["post"], _ratios(g))
    assert_allclose(acc, g["acc"], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("r", [0, 1])
def test_supervised_train(r):
    """MultitrackHmm.supervisedTrain (hmm.py:174-210, emission.py:293-330) on two tables, intervals that
    abut, straddle table ends and fall between tables; with and without segment ratios."""
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    from tehmm_amd.track import TrackData
    g = load_golden("supervised_r%d" % r)
    tabs = [_table(g["obs%d" % i], "chrA", int(g["starts"][i]), g["seg_lens%d" % i]) for i in range(2)]
    beds = [("chrA", int(a), int(b), int(s)) for a, b, s in zip(g["bed_start"], g["bed_end"], g["bed_state"])]
    em = IndependentMultinomialEmissionModel(4, g["symbols"].tolist(), fudge=float(g["fudge"]),
                                             effectiveSegmentLength=(int(g["eff_len"]) if r else None))
    h = MultitrackHmm(em, fudge=float(g["fudge"]))
    h.supervisedTrain(TrackData(tabs, None, g["symbols"].tolist()), beds)
    assert_allclose(h.transmat_, g["transmat"], rtol=1e-12)
    assert_allclose(h._log_transmat, g["log_transmat"], rtol=1e-12)
    assert_allclose(h.startprob_, g["startprob"], rtol=1e-12)
    assert_allclose(h.emissionModel.logProbs, g["log_probs"], rtol=1e-11, atol=1e-13)


# ------------------------------------------------------------------ 8(f) rank 2 + rank 4
def _gauss_tracks(g, values):
    from tehmm_amd.track import CategoryMap, Track, TrackList
    gmap = CategoryMap(reserved=1, defaultVal="0", scale=float(g["gauss_scale"]))
    for v in values:
        gmap.getMap(v, update=True)
    gmap.sort()
    return TrackList([Track("cat", 0), Track("gauss", 1, dist="gaussian", valueMap=gmap), Track("cat2", 2)]), gmap


def _raw_values(g):
    mb = g["mapback"]
    return sorted(set(float(v) for v in mb[1:] if np.isfinite(v)))


def test_segment_plain():
    """TrackTable.segment (mode / gaussian-mean interpolation + compression) against the reference."""
    from tehmm_amd.track import IntegerTrackTable
    g = load_golden("segment_plain")
    tracks, gmap = _gauss_tracks(g, _raw_values(g))
    assert_array_equal(np.where(gmap.getMapBackTable(np.uint8) > 1e300, np.nan, gmap.getMapBackTable(np.uint8)),
                       g["mapback"])
    data, lens, start = g["data"], g["seg_lens"], int(g["start"])
    seg_start = start + np.concatenate([[0], np.cumsum(lens)[:-1]])
    segIntervals = [("chrS", int(a), int(a + l)) for a, l in zip(seg_start, lens)]
    tab = IntegerTrackTable(3, "chrS", start, start + data.shape[0])
    tab.data = data.copy()
    tab.segment(segIntervals, tracks, interpolate=True)
    assert_array_equal(tab.getSegmentOffsets(), g["out_offsets"])
    assert_array_equal(tab.data, g["out_data"])           # modes AND the gaussian symbols (same map)
    assert len(gmap) == int(g["n_symbols_after"])
    assert_array_equal(tab.getSegmentLengthsAsRatio(10.0), g["ratios"])


def test_segment_masked_and_states_to_bed(tmp_path):
    """setMaskTable / getMaskRunningOffsets (_track.runSum), segment on the masked table, and the per-row
    lines of teHmmEval's statesToBed (coordinates through segments and mask, posterior column)."""
    from tehmm_amd import output
    from tehmm_amd.track import IntegerTrackTable
    g = load_golden("segment_masked")
    tracks, gmap = _gauss_tracks(g, _raw_values(g))
    data, lens, start = g["data"], g["seg_lens"], int(g["start"])
    T = data.shape[0]
    mtab = IntegerTrackTable(2, "chrS", start, start + T)
    mtab.data = g["mask"].copy()
    tab = IntegerTrackTable(3, "chrS", start, start + T)
    tab.data = data.copy()
    tab.setMaskTable(mtab)
    assert_array_equal(tab.maskArray.astype(np.uint8), g["keep"])
    assert_array_equal(tab.getMaskRunningOffsets(), g["run_masked"])
    assert_array_equal(tab.getMaskRunningOffsets(reverseTransform=True), g["run_full"])
    assert_array_equal(tab.data, g["data_masked"])
    seg_start = start + np.concatenate([[0], np.cumsum(lens)[:-1]])
    tab.segment([("chrS", int(a), int(a + l)) for a, l in zip(seg_start, lens)], tracks, interpolate=True)
    assert_array_equal(tab.getSegmentOffsets(), g["out_offsets"])
    assert_array_equal(tab.data, g["out_data"])
    assert tab.getEnd() == int(g["table_end"])
    # statesToBed
    psum = (g["post"] * g["post_mask"]).sum(axis=1)
    bed, pd = tmp_path / "out.bed", tmp_path / "pd.bed"
    output.statesToBed(tab, g["states"], str(bed), psum, str(pd), append=False)
    assert bed.read_bytes() == bytes(g["bed_text"])       # byte-identical
    ref_lines = bytes(g["post_text"]).decode().strip().split("\n")
    got_lines = pd.read_text().strip().split("\n")
    assert len(ref_lines) == len(got_lines)
    for a, b in zip(ref_lines, got_lines):
        ca, cb = a.split("\t"), b.split("\t")
        assert ca[:3] == cb[:3]
        # (the fixture was written under Python 3, which prints 17 digits; the reference's Python 2
        #  prints 12 significant digits, which is what the native writer does)
        assert_allclose(float(cb[3]), float(ca[3]), rtol=1e-11)
        assert cb[3] == ("%.12g" % float(ca[3]) if any(c in "%.12g" % float(ca[3]) for c in ".en")
                         else "%.12g" % float(ca[3]) + ".0")


def test_posterior_masksum():
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    model = synth.make_model(35, seed=2)
    lens = [900, 1, 4000, 257]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=4, missing=0.02)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    hm.eval(hb, viterbi=False, posterior=True)
    post = hb.posteriors()
    mask = (np.arange(35) % 4 == 1).astype(np.float64)
    assert_allclose(hb.posterior_masksum(mask), (post * mask).sum(axis=1), rtol=1e-12)
    assert_allclose(hb.posterior_masksum(mask, 100, 1000), (post[100:1000] * mask).sum(axis=1), rtol=1e-12)


# ------------------------------------------------------------------ 8(f) rank 1: device-resident EM
def _golden_hmm(g, eff_len=None, **kw):
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    em = IndependentMultinomialEmissionModel(g["transmat"].shape[0], g["symbols"].tolist(),
                                             effectiveSegmentLength=eff_len)
    em.logProbs = g["log_probs"].copy()
    h = MultitrackHmm(em, **kw)
    h.transmat_ = g["transmat"].copy()
    h.startprob_ = g["startprob"].copy()
    return h, em


@pytest.mark.parametrize("with_ratio", [0, 1])
def test_device_em_matches_reference(with_ratio, monkeypatch):
    """One Baum-Welch iteration entirely on the device (E-step statistics left in HBM, tehmm_model_mstep)
    against the parameters the real reference has after its M-step, and against the host path."""
    g = load_golden("em_iteration_r%d" % with_ratio)
    seqs = [g["obs%d" % i] for i in range(3)]
    tabs = [_table(s, "chrS", 0, g["seg_lens%d" % i]) for i, s in enumerate(seqs)] if with_ratio else seqs
    out = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("TEHMM_DEVICE_EM", dev)
        h, _ = _golden_hmm(g, eff_len=(int(g["eff_len"]) if with_ratio else None), n_iter=2, thresh=0.0,
                           fixStart=False, fudge=0.0)
        h.init_params = ""
        h.fit(tabs)
        assert_allclose(h.transmat_, g["transmat_after"], rtol=RTOL)
        assert_allclose(h.startprob_, g["startprob_after"], rtol=RTOL)
        assert_allclose(h.emissionModel.logProbs, g["log_probs_after"], rtol=RTOL, atol=1e-9)
        assert_allclose(h.last_forward_log_prob, g["last_logprob"], rtol=1e-9)
        out[dev] = h
    assert_allclose(out["1"]._log_transmat, out["0"]._log_transmat, rtol=1e-9)


def test_device_em_gaussian_refit():
    """M-step with a gaussian track (emission.maximize + makeGaussian, quirk Q19) on the device."""
    from tehmm_amd.emission import IndependentMultinomialAndGaussianEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    from tehmm_amd.track import CategoryMap, Track, TrackList
    g = load_golden("mstep_gauss")
    gk = int(g["gauss_track"])
    gmap = CategoryMap(reserved=1, defaultVal="0", scale=0.5)
    for v in range(0, 60, 2):
        gmap.getMap(v, update=True)
    gmap.sort()
    vals = np.asarray([gmap.getMapBack(s) for s in range(1, len(g["gauss_values"]))], dtype=np.float64)
    assert_array_equal(vals, g["gauss_values"][1:])
    tracks = TrackList([Track("cat", 0), Track("gauss", 1, dist="gaussian", valueMap=gmap), Track("cat2", 2)])
    N = g["transmat"].shape[0]
    em = IndependentMultinomialAndGaussianEmissionModel(N, g["symbols"].tolist(), tracks, fudge=0.0)
    em.logProbs = g["log_probs"].copy()
    h = MultitrackHmm(em, n_iter=2, thresh=0.0, fixStart=False)
    h.transmat_ = g["transmat"].copy()
    h.startprob_ = np.full(N, 1.0 / N)
    h.init_params = ""
    h.trackList = tracks
    assert h._can_fit_on_device([g["obs0"], g["obs1"]])
    h.fit([g["obs0"], g["obs1"]])
    assert_allclose(h.transmat_, g["transmat_after"], rtol=RTOL)
    assert_allclose(h.startprob_, g["startprob_after"], rtol=RTOL)
    assert_allclose(h.emissionModel.gaussParams[gk], g["gauss_params_after"][gk], rtol=RTOL)
    assert_allclose(h.emissionModel.logProbs, g["log_probs_after"], rtol=RTOL, atol=1e-9)
    assert_allclose(h.last_forward_log_prob, g["last_logprob"], rtol=1e-9)


@pytest.mark.parametrize("device_em", ["1", "0"])
def test_em_maxprob_bookkeeping(device_em, monkeypatch):
    """--maxProb (hmm.py:690-711, quirk Q17): the best-iteration copy follows the reference's
    per-sequence bookkeeping -- Python-2 ordering of None included -- on the fused, the device-resident
    and the hook paths, which must agree with each other."""
    from tehmm_amd import synth
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    from tehmm_amd.track import TrackData
    monkeypatch.setenv("TEHMM_DEVICE_EM", device_em)
    model = synth.make_model(6, (3, 5, 4), (), seed=21)
    start = synth.make_model(6, (3, 5, 4), (), seed=22)
    seqs = [synth.sample_obs(model, T, seed=30 + i, missing=0.03) for i, T in enumerate((400, 150, 300))]
    res = {}
    for path in ("fused", "hooks"):
        em = IndependentMultinomialEmissionModel(6, [3, 5, 4])
        em.logProbs = start.log_probs.copy()
        h = MultitrackHmm(em, n_iter=4, thresh=0.0, maxProb=True, fixStart=False)
        h.transmat_ = start.transmat.copy()
        h.init_params = ""
        if path == "hooks":
            h._can_fuse = lambda tables: False            # BaseHMM._do_estep over the array-level hooks
        h.train(TrackData(seqs, None, [3, 5, 4]))
        assert h.bestCopy is not None and h.best_forward_log_prob is not None
        res[path] = h
    a, b = res["fused"], res["hooks"]
    assert a.bestCopy.current_iteration == b.bestCopy.current_iteration
    assert_allclose(a.best_forward_log_prob, b.best_forward_log_prob, rtol=1e-9)
    assert_allclose(a.last_forward_log_prob, b.last_forward_log_prob, rtol=1e-9)
    assert_allclose(a.transmat_, b.transmat_, rtol=1e-6)
    assert_allclose(a.emissionModel.logProbs, b.emissionModel.logProbs, rtol=1e-6, atol=1e-9)
    # the trained model is NOT the untrained start (the bug ADVICE r1 describes)
    assert np.abs(a.transmat_ - start.transmat).max() > 1e-3


# ------------------------------------------------------------------ full-size configurations
def _tiled_obs(model, T, seed, piece_len=50_000, noise_p=0.2):
    from tehmm_amd import synth
    rs = np.random.RandomState(seed)
    piece = synth.sample_obs(model, piece_len, seed=seed + 1)
    obs = np.tile(piece, ((T + piece_len - 1) // piece_len, 1))[:T].copy()
    noise = rs.rand(T) < noise_p
    for k, sk in enumerate(model.symbols_per_track):
        obs[noise, k] = rs.randint(1, sk + 1, size=int(noise.sum()))
    return obs


def test_config2_full_size_vs_oracle():
    """BASELINE configs[1] at its full size, default knobs: ONE 10 Mb interval, 35 states / 10 tracks.
    Viterbi path and score bit-exact against the CPU oracle (binades up to 2^27, where the quantised
    integer max-plus of the chunk-parallel pass is most fragile), forward log-likelihood at 1e-6."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(35, seed=0)
    T = 10_000_000
    obs = _tiled_obs(model, T, seed=77)
    offs = np.asarray([0, T], dtype=np.int64)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    res = hm.eval(hb, viterbi=True, posterior=True)
    paths = hb.paths()
    tm = hb.timing()
    assert tm.get("count:viterbi_chunk_jumps", 0) > 0          # the chunk-parallel path really ran
    rowsum_dev = hb.posterior_masksum(np.ones(35))
    assert_allclose(rowsum_dev, 1.0, rtol=1e-9)
    hb.close()
    # the oracle's two sequential passes over 10 Mb side by side (ctypes calls release the GIL): 35 s instead of 50
    import threading
    out = {}

    def _forward():
        frame = oracle.emission(obs, model.log_probs, 1.0, None)
        fwd = oracle.forward(model.log_startprob, model.log_transmat, frame)
        out["flp"] = oracle.logsumexp(fwd[-1])

    th = threading.Thread(target=_forward)
    th.start()
    lp_o, path_o = oracle.decode(obs, model.log_probs, model.log_startprob, model.log_transmat)
    th.join()
    assert_array_equal(paths, path_o)
    assert res["viterbi_logprob"][0] == lp_o
    del path_o, paths
    assert_allclose(res["forward_logprob"][0], out["flp"], rtol=RTOL)


@pytest.mark.parametrize("variant", ["sticky", "sparse"])
def test_long_interval_model_variants_vs_oracle(variant):
    """3 Mb single interval on the models the speculation likes least: a sticky chain (self-transition
    0.995, like trained TE models) and sparse (-1e100) transitions.  Bit-exact Viterbi vs the oracle."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    kw = dict(stay=0.995) if variant == "sticky" else dict(sparse=0.5)
    model = synth.make_model(35, seed=4, **kw)
    T = 3_000_000
    obs = _tiled_obs(model, T, seed=78)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, np.asarray([0, T], dtype=np.int64))
    res = hm.eval(hb, viterbi=True, posterior=False)
    paths = hb.paths()
    hb.close()
    lp_o, path_o = oracle.decode(obs, model.log_probs, model.log_startprob, model.log_transmat)
    assert_array_equal(paths, path_o)
    assert res["viterbi_logprob"][0] == lp_o


SPEC_ENV = ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0",
            "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED")


def _mixed_ratios(T, seed, big=False):
    """Segment ratios as a segmented table yields them: runs of 1.0 (unsegmented stretches), values above
    and below 1, and -- big -- segments long enough that lt[j][j] * (r - 1) moves the score by hundreds."""
    from tehmm_amd import synth
    rs = np.random.RandomState(seed)
    r = synth.random_ratios(T, seed=seed)
    r[rs.rand(T) < 0.4] = 1.0
    if big:
        idx = rs.randint(0, T, size=max(1, T // 500))
        r[idx] = rs.randint(200, 5000, size=idx.size).astype(np.float64) / 20.0
    return np.ascontiguousarray(r)


@pytest.mark.parametrize("N,env,big", [
    (35, {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"}, False),
    (35, {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64", "TEHMM_VIT_RUNS": "0"}, True),
    (35, {"TEHMM_SPEC_CHUNK": "1024"}, True),
    (20, {"TEHMM_SPEC_CHUNK": "512", "TEHMM_LANE_SUB": "128"}, False),
    (7, {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"}, True),
    (3, {"TEHMM_SPEC_CHUNK": "256"}, False),                 # one output group per row (NT = 4)
    (2, {"TEHMM_SPEC_CHUNK": "512", "TEHMM_LANE_SUB": "256"}, True),
])
def test_chunk_parallel_viterbi_with_segment_ratios(monkeypatch, N, env, big):
    """Decode on a segmented table (_hmm.pyx:229-247: lt[j][j] * (r - 1) for r > 1 on every candidate but the
    from-state-0 one, which gets lt[j][j] * r, quirk Q4) through the chunk-parallel exact Viterbi path: paths and
    scores bit-exact against the oracle, and chunks really are jumped over."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in SPEC_ENV:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model = synth.make_model(N, seed=11 + N)
    lens = [1, 300, 5000, 60000, 150000] if env["TEHMM_SPEC_CHUNK"] != "1024" else [700, 90000, 400000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=N, missing=0.03)
    ratios = _mixed_ratios(int(offs[-1]), seed=N + len(env), big=big)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    for want_post in (False, True):             # Viterbi alone, then next to the posterior pipeline
        res = hm.eval(hb, viterbi=True, posterior=want_post)
        t = hb.timing()
        assert t["count:viterbi_chunk_jumps"] > 0, t
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                  model.log_transmat, 1.0, ratios, n_threads=4)
    assert_array_equal(hb.paths(), p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
    assert_allclose(hb.posteriors(), post_o, rtol=RTOL, atol=1e-15)
    hb.close()


def test_segment_ratio_self_transition_zero_uses_sequential_path(monkeypatch):
    """A state that cannot follow itself (lt[j][j] = -inf) makes the ratio products -inf / NaN: such models stay
    on the sequential kernels, whose arithmetic is the reference's own."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in SPEC_ENV:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TEHMM_SPEC_CHUNK", "256")
    monkeypatch.setenv("TEHMM_LANE_SUB", "64")
    model = synth.make_model(12, seed=5)
    lt = model.log_transmat.copy()
    lt[3, 3] = -np.inf
    lens = [4000, 30000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=1)
    ratios = _mixed_ratios(int(offs[-1]), seed=3)
    ratios[ratios < 1.0] = 1.0                    # (0 * -inf would be NaN in the reference too)
    hm = HipModel(lt, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, viterbi=True, posterior=False)
    assert "count:viterbi_chunk_jumps" not in hb.timing()
    p_o, vlp_o, _, _ = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob, lt, 1.0, ratios,
                                         want_post=False, n_threads=4)
    assert_array_equal(hb.paths(), p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    hb.close()


def test_config3b_k32_vs_oracle():
    """BASELINE configs[2] variant 3b: the alyrata track-XML shape, K = 32 (15 multinomial + 14 gaussian /
    250 bins + 3 binary), several intervals incl. ragged ones: Viterbi bit-exact, posteriors 1e-6."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(35, synth.CONFIG3B_SYMBOLS, synth.CONFIG3B_GAUSSIAN, seed=9)
    lens = [300_000, 1, 70_001, 2049, 150_000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = _tiled_obs(model, int(offs[-1]), seed=79, piece_len=20_000)
    obs[np.random.RandomState(1).rand(*obs.shape) < 0.02] = 0
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    res = hm.eval(hb, viterbi=True, posterior=True)
    paths, post = hb.paths(), hb.posteriors()
    hb.close()
    rp, rv, rf, rpost = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob, model.log_transmat,
                                          n_threads=8)
    assert_array_equal(paths, rp)
    assert_array_equal(res["viterbi_logprob"], rv)
    assert_allclose(res["forward_logprob"], rf, rtol=RTOL)
    assert_allclose(post, rpost, rtol=RTOL, atol=1e-12)


def test_config4_estep_100kb_chunks_vs_oracle():
    """BASELINE configs[3] geometry: 35 states, K = 12 (10 multinomial + 2 gaussian), 100 kb training
    chunks.  Statistics of one E-step over 12 chunks against the oracle's per-sequence E-step."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(35, synth.CONFIG4_SYMBOLS, synth.CONFIG4_GAUSSIAN, seed=12)
    n, L = 12, 100_000
    offs = (np.arange(n + 1) * L).astype(np.int64)
    obs = _tiled_obs(model, n * L, seed=80, piece_len=25_000)
    K, N, S = model.log_probs.shape
    start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    lp = hm.estep(hb, False, start, trans, st)
    hb.close()
    ref = oracle.estep([obs[offs[i]:offs[i + 1]] for i in range(n)], model.log_probs, model.log_startprob,
                       model.log_transmat, 1.0, None)
    assert_allclose(lp, ref["logprob"], rtol=RTOL)
    assert_allclose(start, ref["start"], rtol=RTOL, atol=1e-12)
    assert_allclose(trans, ref["trans"], rtol=RTOL, atol=1e-9)
    assert_allclose(st, ref["obs"], rtol=RTOL, atol=1e-9)


def test_out_of_range_symbols_match_array_level():
    """A symbol beyond a track's last one reads the reference's zero padding (emission.py:136-138): the
    fused path, the array-level path and the oracle agree."""
    from tehmm_amd import _emission, synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(12, (3, 4, 9), (), seed=5)
    T = 3000
    obs = synth.sample_obs(model, T, seed=6)
    rs = np.random.RandomState(7)
    bad = rs.rand(T) < 0.05
    obs[bad, 0] = rs.randint(4, 10, size=int(bad.sum()))          # track 0 has symbols 1..3; table width 10
    frame = np.zeros((T, 12))
    _emission.fastAllLogProbs(obs, model.log_probs, frame, 1.0, None)
    assert_array_equal(frame, oracle.emission(obs, model.log_probs, 1.0, None))
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, np.asarray([0, T], dtype=np.int64))
    res = hm.eval(hb, viterbi=True, posterior=True)
    lp_o, path_o = oracle.decode(obs, model.log_probs, model.log_startprob, model.log_transmat)
    assert_array_equal(hb.paths(), path_o)
    assert res["viterbi_logprob"][0] == lp_o
    flp, post = oracle.score_samples(obs, model.log_probs, model.log_startprob, model.log_transmat)
    assert_allclose(res["forward_logprob"][0], flp, rtol=1e-9)
    assert_allclose(hb.posteriors(), post, rtol=RTOL, atol=1e-14)


def test_eval_tables_empty():
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    h = MultitrackHmm(IndependentMultinomialEmissionModel(3, [2]))
    out = h._eval_tables([], viterbi=True, posterior=True)
    assert out["paths"] == [] and out["posteriors"] == []


def test_rccl_single_rank_allreduce(tmp_path):
    """The RCCL path with one rank (nccl backend on the one GPU of this box): allreduce_stats, the
    device-statistics all-reduce of the EM loop and the uint8 path all-gather; record kept under
    gpurun_out/."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text('''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from tehmm_amd import _lib, synth, dist as tdist
from tehmm_amd.engine import HipModel, HipBatch, DeviceStats
torch.cuda.set_device(0)
_lib.check(_lib.load().tehmm_set_device(0))
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
model = synth.make_model(35, seed=0)
obs = synth.sample_obs(model, 6000, seed=1)
offs = np.asarray([0, 2500, 6000], dtype=np.int64)
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
hb = HipBatch(obs, offs)
st = DeviceStats(hm)
assert st.tensor is not None
lp = hm.estep_device(hb, False, st)
before = st.tensor.clone()
tdist.allreduce_device_stats(st)          # world size 1: returns early
dist.all_reduce(st.tensor)                # the collective itself, over RCCL
assert torch.equal(before, st.tensor)
stats = {"nobs": 2, "start": np.ones(35), "trans": np.ones((35, 35)), "obs": np.ones((10, 35, 251))}
t = torch.from_numpy(tdist.pack_stats(stats, -1.5)).cuda()
dist.all_reduce(t)
res = hm.eval(hb, viterbi=True, posterior=False)
paths = hb.paths()
send = torch.from_numpy(paths.astype(np.uint8)).cuda()
recv = torch.empty_like(send)
dist.all_gather_into_tensor(recv, send)
assert np.array_equal(recv.cpu().numpy(), paths.astype(np.uint8))
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "logprob": lp,
                  "stats_doubles": int(st.size), "packed_doubles": int(t.numel()), "ok": True}))
dist.destroy_process_group()
''' % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    outdir = os.path.join(root, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "rccl_single_rank.json"), "w") as f:
        f.write(line + "\n")
    assert '"ok": true' in line
