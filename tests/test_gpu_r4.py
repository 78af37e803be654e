"""Round-4 GPU tests (through the C ABI): the three-wave quantised Viterbi pass and emission + P0 kernels against the
oracle and against the one-wave kernels (even and uneven splits of the output groups, segment ratios), binade placement
on the device against the host placement, bit-reproducible E-step statistics, the count-static fused passes on ragged
intervals."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip():
    from tehmm_amd import _lib, build
    build.build()
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path has no CPU fallback")


KNOBS = ("TEHMM_P2_SPLIT", "TEHMM_EMIS_SPLIT", "TEHMM_DEVICE_PLACE", "TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_DEFER",
         "TEHMM_LANE_VIT", "TEHMM_ESTEP_FUSED", "TEHMM_SOFT_TIES")


def _noisy_obs(model, total, seed):
    from tehmm_amd import synth
    piece = synth.sample_obs(model, 40_000, seed=seed)
    obs = np.tile(piece, (total // 40_000 + 1, 1))[:total].copy()
    rs = np.random.RandomState(seed + 1)
    noise = rs.rand(total) < 0.3
    for k, sk in enumerate(model.symbols_per_track):
        obs[noise, k] = rs.randint(1, sk + 1, size=int(noise.sum()))
    return obs


def _eval(model, obs, offs, ratios=None, **kw):
    from tehmm_amd.engine import HipBatch, HipModel
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, use_ratios=ratios is not None, **kw)
    out = (hb.paths() if kw.get("viterbi") else None, res.get("viterbi_logprob"),
           hb.posteriors() if kw.get("posterior") else None, res.get("forward_logprob"), hb.timing())
    hb.close()
    hm.close()
    return out


# N -> padded states / output groups: 9 -> 12 / 3 (one group per wave), 13 -> 16 / 4 (2, 2, 0: a wave without outputs),
# 35 -> 36 / 9 (even), 38 -> 40 / 10 (4, 4, 2), 50 -> 56 / 14 (5, 5, 4), 60 -> 64 / 16 (6, 6, 4)
@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,ratio", [(9, False), (13, False), (35, False), (35, True), (38, False), (50, True), (60, False)])
def test_three_wave_passes_bit_exact(monkeypatch, N, ratio):
    """k_vit_lane3 (quantised pass split over three waves) and k_emis_gain_lane3 against the oracle, bit for bit, and
    against the one-wave kernels: same paths, same scores, the same number of exact blocks left to the chain."""
    from oracle import oracle
    from tehmm_amd import synth
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(N, (3, 5, 4, 30, 250) if N != 9 else (3, 5, 4), (4,) if N != 9 else (), seed=40 + N)
    lens = [180_000, 70_001, 3_000, 1]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = _noisy_obs(model, int(offs[-1]), seed=N)
    ratios = None
    if ratio:
        rs = np.random.RandomState(N)
        ratios = np.minimum(1 + rs.geometric(1 / 20.0, size=int(offs[-1])), 100) / 20.0
    got = {}
    monkeypatch.setenv("TEHMM_SOFT_TIES", "0")             # (the one-wave kernel ends a piece at every rounding tie)
    for tag, p2, em in (("split", "1", "1"), ("one", "0", "0")):
        monkeypatch.setenv("TEHMM_P2_SPLIT", p2)
        monkeypatch.setenv("TEHMM_EMIS_SPLIT", em)
        got[tag] = _eval(model, obs, offs, ratios, viterbi=True, posterior=False)
    for i in range(len(lens)):
        sl = slice(int(offs[i]), int(offs[i + 1]))
        vlp, path = oracle.decode(obs[sl], model.log_probs, model.log_startprob, model.log_transmat, 1.0,
                                  None if ratios is None else ratios[sl])
        for tag in ("split", "one"):
            assert_array_equal(got[tag][0][sl], path)
            assert got[tag][1][i] == vlp
    ts, to = got["split"][4], got["one"][4]
    assert ts["count:viterbi_chunk_jumps"] > 0
    assert ts["count:viterbi_exact_blocks"] == to["count:viterbi_exact_blocks"]
    assert ts["count:viterbi_chunk_jumps"] == to["count:viterbi_chunk_jumps"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,chunk", [(35, None), (35, "256"), (20, "128"), (50, None), (63, "256")])
def test_soft_ties_bit_exact_and_fewer_exact_blocks(monkeypatch, N, chunk):
    """Soft ties (round 4): the quantised pass goes through rounding ties on the even-delta hypothesis and the exact
    chain passes them when its verified delta has the matching parity.  Paths and scores bit for bit against the oracle
    -- default and small chunks (every tie, link and parity case at sizes the oracle checks in seconds), low binades where
    ties come every few dozen positions -- and fewer exact blocks than with every tie ending a piece."""
    from oracle import oracle
    from tehmm_amd import synth
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    if chunk is not None:
        monkeypatch.setenv("TEHMM_SPEC_CHUNK", chunk)
        monkeypatch.setenv("TEHMM_LANE_SUB", "128" if int(chunk) >= 128 else chunk)
    model = synth.make_model(N, (3, 5, 4, 30, 250), (4,), seed=70 + N)
    lens = [260_000, 90_001, 40_000, 5_000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = _noisy_obs(model, int(offs[-1]), seed=N + 5)
    got = {}
    for soft in ("1", "0"):
        monkeypatch.setenv("TEHMM_SOFT_TIES", soft)
        got[soft] = _eval(model, obs, offs, viterbi=True, posterior=False)
    for i in range(len(lens)):
        sl = slice(int(offs[i]), int(offs[i + 1]))
        vlp, path = oracle.decode(obs[sl], model.log_probs, model.log_startprob, model.log_transmat)
        for soft in ("1", "0"):
            assert_array_equal(got[soft][0][sl], path)
            assert got[soft][1][i] == vlp
    assert got["1"][4]["count:viterbi_chunk_jumps"] > 0
    assert got["1"][4]["count:viterbi_exact_blocks"] < got["0"][4]["count:viterbi_exact_blocks"]


@pytest.mark.timeout(900)
def test_device_placement_matches_host_placement(monkeypatch):
    """Binade placement and the work list of the quantised pass built on the device (tehmm_place.hip.h) against the
    host path of rounds 1..3: same paths and scores as the oracle, the same chunks speculated and jumped."""
    from oracle import oracle
    from tehmm_amd import synth
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(35, seed=0)
    lens = [400_000, 150_000, 64 * 1024 + 5, 2_000, 1]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = _noisy_obs(model, int(offs[-1]), seed=77)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TEHMM_DEVICE_PLACE", mode)
        got[mode] = _eval(model, obs, offs, viterbi=True, posterior=True)
    for i in range(len(lens)):
        sl = slice(int(offs[i]), int(offs[i + 1]))
        vlp, path = oracle.decode(obs[sl], model.log_probs, model.log_startprob, model.log_transmat)
        for mode in ("1", "0"):
            assert_array_equal(got[mode][0][sl], path)
            assert got[mode][1][i] == vlp
    assert got["1"][4]["count:viterbi_exact_blocks"] == got["0"][4]["count:viterbi_exact_blocks"]
    assert got["1"][4]["count:viterbi_chunk_jumps"] == got["0"][4]["count:viterbi_chunk_jumps"] > 0
    assert_array_equal(got["1"][2], got["0"][2])          # (the posterior pipeline does not depend on the placement)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,symbols,gauss", [(35, None, None), (50, (4, 250, 30, 250, 250), (1, 3, 4))])
def test_estep_statistics_bit_reproducible(monkeypatch, N, symbols, gauss):
    """Same input -> same bits (round 4): the fused E-step three times on fresh batch handles and twice per handle gives
    IDENTICAL statistics and log-likelihood -- the reductions fold per-writer partial sums in a fixed order and the LDS
    histograms are fixed-point integers -- and they agree with the oracle at 1e-6."""
    from oracle import oracle
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    model = (synth.make_model(N, synth.CONFIG4_SYMBOLS, synth.CONFIG4_GAUSSIAN, seed=12) if symbols is None
             else synth.make_model(N, symbols, gauss, seed=12 + N))
    lens = [100_000] * 3 + [33_333, 1, 2_500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
    K, _, S = model.log_probs.shape
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    runs = []
    for _rep in range(3):
        hb = HipBatch(obs, offs)
        for _again in range(2):
            start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
            lp = hm.estep(hb, False, start, trans, st)
            runs.append((lp, start, trans, st))
        assert "estep_reduce" in hb.timing()               # (the chunk-parallel path ran)
        hb.close()
    hm.close()
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        assert_array_equal(r[1], runs[0][1])
        assert_array_equal(r[2], runs[0][2])
        assert_array_equal(r[3], runs[0][3])
    ref = oracle.estep([obs[offs[i]:offs[i + 1]] for i in range(len(lens))], model.log_probs, model.log_startprob,
                       model.log_transmat, 1.0, None)
    assert_allclose(runs[0][0], ref["logprob"], rtol=1e-9)
    assert_allclose(runs[0][2], ref["trans"], rtol=1e-6, atol=1e-9)
    assert_allclose(runs[0][3], ref["obs"], rtol=1e-6, atol=1e-9)


@pytest.mark.timeout(900)
def test_fused_passes_ragged_geometry_vs_oracle(monkeypatch):
    """The fused forward / backward passes store unconditionally since round 4 (warm-up steps and lanes whose item is
    not run write rows that are overwritten or never read): ragged intervals -- tails shorter than a chunk, one-row and
    sub-chunk intervals, a last group with empty lanes -- with small chunks so that every kind of item occurs; posteriors
    and log-likelihoods against the oracle at 1e-6, and a second evaluation of the same batch gives the same bits."""
    from oracle import oracle
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TEHMM_SPEC_CHUNK", "256")
    monkeypatch.setenv("TEHMM_LANE_SUB", "128")
    model = synth.make_model(35, seed=3)
    lens = [9_000, 1, 255, 256, 257, 4_097, 700, 12_345]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=9, missing=0.05)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs, offs)
    res = hm.eval(hb, viterbi=True, posterior=True)
    post = hb.posteriors().copy()
    paths = hb.paths().copy()
    res2 = hm.eval(hb, viterbi=True, posterior=True)
    assert_array_equal(hb.posteriors(), post)
    assert_array_equal(hb.paths(), paths)
    assert_array_equal(res2["forward_logprob"], res["forward_logprob"])
    hb.close()
    hm.close()
    ref_paths, ref_vlp, ref_flp, ref_post = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                               model.log_transmat)
    assert_array_equal(paths, ref_paths)
    assert_array_equal(res["viterbi_logprob"], ref_vlp)
    assert_allclose(res["forward_logprob"], ref_flp, rtol=1e-9)
    assert_allclose(post, ref_post, rtol=1e-6, atol=1e-12)


@pytest.mark.timeout(900)
def test_eval_stream_matches_one_batch(monkeypatch):
    """engine.eval_stream (interval groups, evaluation of group g + 1 on a worker thread while group g's results cross
    PCIe) against one evaluation of the whole batch: same paths and scores; posteriors and masked sums at 1e-6 (the groups
    are small batches that take other kernels than the whole batch: both are within 1e-6 of the reference)."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel, eval_stream
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(35, seed=5)
    lens = [30_000, 1, 12_345, 50_000, 700, 41_000, 8_192]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=21, missing=0.03)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs, offs)
    res = hm.eval(hb, viterbi=True, posterior=True)
    paths, post = hb.paths().copy(), hb.posteriors().copy()
    hb.close()
    ps, qs, vlp, flp = eval_stream(hm, obs, offs, group_rows=45_000)
    mask = (np.arange(35) % 3 == 0).astype(np.float64)
    _, ms, _, _ = eval_stream(hm, obs, offs, group_rows=45_000, viterbi=False, mask=mask)
    hm.close()
    assert_array_equal(vlp, res["viterbi_logprob"])
    assert_allclose(flp, res["forward_logprob"], rtol=1e-9)
    for i in range(len(lens)):
        sl = slice(int(offs[i]), int(offs[i + 1]))
        assert_array_equal(ps[i], paths[sl])
        assert_allclose(qs[i], post[sl], rtol=1e-6, atol=1e-15)
        assert_allclose(ms[i], post[sl] @ mask, rtol=1e-6, atol=1e-15)


# ------------------------------------------------------------------ E-step on the item-parallel passes (64..128 states, ratios)
def _rel(a, b, floor=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.abs(b) > floor
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


def _seg_ratios(total, seed, mean=4.0):
    """Segment-length ratios as emission.getSegmentRatios gives them: length / mean length, many of them 1."""
    rs = np.random.RandomState(seed)
    r = np.clip(rs.geometric(1.0 / mean, size=total), 1, 60).astype(np.float64) / mean
    r[rs.rand(total) < 0.3] = 1.0
    return r


@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,symbols,gauss,use_ratios", [
    (100, (3, 5, 4, 30), (), False),                      # BASELINE configs[4] model
    (100, (3, 5, 4, 30), (), True),                       # ... trained on a segmented table
    (128, (2, 7), (), True),                              # the largest model, no pad states
    (64, (3, 5, 4, 30), (), False),                       # the smallest one the fused passes do not take
    (35, (3, 5, 4, 30, 250), (4,), True),                 # segment ratios below 64 states: 48 padded states here
    (5, (3, 5, 4), (), True),                             # one state tile
    (60, (4, 250, 30), (1,), True),                       # 64 padded states, a 250-bin track in the gamma product
    (20, (2,) * 20, (), True),                            # 32 padded states; 20 tracks: observation words beyond the fourth
])
def test_wide_estep_vs_oracle(monkeypatch, N, symbols, gauss, use_ratios):
    """tehmm_estep_batch on the item-parallel passes (k_wide_emis_tile, k_wide_fwd, k_wide_bwd<ESTEP>, k_wide_estep_xi,
    k_wide_estep_rows) against the oracle's per-sequence E-step (basehmm.py:504-523, hmm.py:545-574,
    _hmm.pyx:62-117 with the diagonal ratio term, _emission.pyx:183-190 with ratio-weighted posteriors) on ragged
    intervals: one-row, sub-item and multi-item intervals, tails shorter than an item.  Statistics at 1e-6 (written
    here); the observed error is asserted at 3e-7 so that a loss of margin shows; same input, same bits."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in KNOBS + ("TEHMM_ESTEP_WIDE", "TEHMM_WIDE_SUB", "TEHMM_LANE_WARMUP"):
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(N, symbols, gauss, seed=12 + N)
    rs = np.random.RandomState(3 + N)
    lens = [int(x) for x in rs.randint(3000, 9000, size=4)] + [1, 70, 1500, 1024, 2048 + 64]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offs[-1])
    obs = synth.sample_obs(model, total, seed=5, missing=0.02)
    r = _seg_ratios(total, 7) if use_ratios else None
    K, _, S = model.log_probs.shape
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    got = []
    for rep in range(2):
        hb = HipBatch(obs, offs, r)
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp = hm.estep(hb, use_ratios, start, trans, st)
        tm = hb.timing()
        got.append((lp, start, trans, st, hb.interval_logprobs()))
        hb.close()
        assert "estep_emission_rows" in tm and tm["count:wide_estep_attempts"] >= 1       # the path under test ran
    ref = oracle.estep([obs[offs[i]:offs[i + 1]] for i in range(len(lens))], model.log_probs, model.log_startprob,
                       model.log_transmat, 1.0, [r[offs[i]:offs[i + 1]] for i in range(len(lens))] if use_ratios else None)
    lp, start, trans, st, ilp = got[0]
    assert_allclose(lp, ref["logprob"], rtol=1e-9)
    assert_allclose(ilp.sum(), ref["logprob"], rtol=1e-9)
    assert_allclose(start, ref["start"], rtol=1e-6, atol=1e-12)
    assert_allclose(trans, ref["trans"], rtol=1e-6, atol=1e-9)
    assert_allclose(st, ref["obs"], rtol=1e-6, atol=1e-9)
    worst = max(_rel(start, ref["start"]), _rel(trans, ref["trans"], 1e-6), _rel(st, ref["obs"], 1e-6))
    print("wide E-step N=%d ratios=%s: max rel error of the statistics %.3g" % (N, use_ratios, worst))
    assert worst <= 3e-7
    for a, b in zip(got[0], got[1]):                                   # reproducible sums (ordered folds)
        assert_array_equal(np.asarray(a), np.asarray(b))
    if N < 64:
        # the sequential kernels (k_fb_coop<TRATIO> + k_estep_accum) on the same batch
        monkeypatch.setenv("TEHMM_ESTEP_WIDE", "0")
        hb = HipBatch(obs, offs, r)
        s2, t2, st2 = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp2 = hm.estep(hb, use_ratios, s2, t2, st2)
        assert "estep_emission_rows" not in hb.timing()
        hb.close()
        assert_allclose(lp2, lp, rtol=1e-9)
        assert_allclose(t2, trans, rtol=1e-6, atol=1e-9)
        assert_allclose(st2, st, rtol=1e-6, atol=1e-9)
    hm.close()


def test_wide_estep_falls_back(monkeypatch):
    """What the item-parallel passes do not take: a row no state can emit (the sequential kernels own the NaN semantics
    below 64 states; TEHMM_ERR_UNSUPPORTED at 64 and above, where MultitrackHmm._do_estep then runs the reference's
    per-sequence loop over the array-level entry points)."""
    from tehmm_amd import _lib, synth
    from tehmm_amd.engine import HipBatch, HipModel
    for k in KNOBS + ("TEHMM_ESTEP_WIDE",):
        monkeypatch.delenv(k, raising=False)
    for N in (6, 70):
        model = synth.make_model(N, (3, 5, 4), (), seed=21)
        lp3 = model.log_probs.copy()
        lp3[1, :, 2] = -np.inf                      # symbol 2 of track 1 cannot be emitted by any state
        lens = [5000, 3000]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        obs = synth.sample_obs(model, int(offs[-1]), seed=2)
        obs[:, 1] = np.where(obs[:, 1] == 2, 1, obs[:, 1])
        obs[6000, 1] = 2
        r = _seg_ratios(int(offs[-1]), 3)
        hm = HipModel(model.log_transmat, model.log_startprob, lp3, 1.0, model.symbols_per_track)
        hb = HipBatch(obs, offs, r)
        K, _, S = lp3.shape
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        if N >= 64:
            with pytest.raises(_lib.TeHmmHipError) as ei:
                hm.estep(hb, True, start, trans, st)
            assert ei.value.code == -3
        else:
            lp = hm.estep(hb, True, start, trans, st)
            assert np.isnan(lp) and np.isnan(trans).all()           # the reference's NaN lattices (sequential path)
        hb.close()
        hm.close()


def test_device_em_falls_back_to_the_host_loop_at_70_states(monkeypatch):
    """MultitrackHmm.fit at 70 states on a table with a row no state can emit: the item-parallel passes refuse it
    (TEHMM_ERR_UNSUPPORTED), _fit_device hands the whole fit to the reference's loop over the array-level entry points
    -- the same parameters as with the device path switched off."""
    from tehmm_amd import synth
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    for k in KNOBS + ("TEHMM_ESTEP_WIDE", "TEHMM_DEVICE_EM"):
        monkeypatch.delenv(k, raising=False)
    N, sym = 70, [3, 5, 4]
    init = synth.make_model(N, tuple(sym), (), seed=4)
    lp3 = init.log_probs.copy()
    lp3[1, :, 2] = -np.inf                       # symbol 2 of track 1: no state can emit it
    lp3[1, :, 1:6] -= np.log(np.exp(lp3[1, :, 1:6]).sum(axis=1))[:, None]     # (still a distribution: validate())
    seqs = [synth.sample_obs(init, T, seed=60 + i) for i, T in enumerate([1500, 900])]
    seqs[0][:, 1] = np.where(seqs[0][:, 1] == 2, 1, seqs[0][:, 1])
    seqs[1][:, 1] = np.where(seqs[1][:, 1] == 2, 1, seqs[1][:, 1])
    seqs[1][0, 1] = 2                            # a LEADING impossible row: the reference skips it (quirk, _emission.pyx:73-80)

    def fresh():
        em = IndependentMultinomialEmissionModel(N, sym)
        em.logProbs = lp3.copy()
        h = MultitrackHmm(em, n_iter=2, thresh=0.0, fixStart=False)
        h.transmat_ = init.transmat.copy()
        h.init_params = ""
        return h

    a = fresh()
    assert a._can_fit_on_device(seqs)
    a.fit(seqs)
    assert a.init_params == ""
    monkeypatch.setenv("TEHMM_DEVICE_EM", "0")
    b = fresh()
    b.fit(seqs)
    assert_allclose(a._log_transmat, b._log_transmat, rtol=1e-9, atol=1e-12)
    assert_allclose(a.emissionModel.logProbs, b.emissionModel.logProbs, rtol=1e-9, atol=1e-12)


@pytest.mark.timeout(900)
def test_device_em_100_states_matches_host_loop(monkeypatch):
    """MultitrackHmm.fit at 100 states: device-resident EM (_fit_device over tehmm_estep_batch_device on the
    item-parallel passes + tehmm_model_mstep) against BaseHMM.fit's per-sequence loop over the array-level entry
    points (TEHMM_DEVICE_EM=0, TEHMM_ESTEP_WIDE=0: what rounds 1-3 ran at this size), two iterations."""
    from tehmm_amd import synth
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    for k in KNOBS + ("TEHMM_ESTEP_WIDE", "TEHMM_DEVICE_EM"):
        monkeypatch.delenv(k, raising=False)
    N, sym = 100, [3, 5, 4, 30]
    truth = synth.make_model(N, tuple(sym), (), seed=3)
    init = synth.make_model(N, tuple(sym), (), seed=4)
    seqs = [synth.sample_obs(truth, T, seed=50 + i, missing=0.02) for i, T in enumerate([2500, 1800, 700])]

    def fresh():
        em = IndependentMultinomialEmissionModel(N, sym)
        em.logProbs = init.log_probs.copy()
        h = MultitrackHmm(em, n_iter=2, thresh=0.0, fixStart=False)
        h.transmat_ = init.transmat.copy()
        h.init_params = ""
        return h

    a = fresh()
    assert a._can_fit_on_device(seqs)
    a.fit(seqs)
    monkeypatch.setenv("TEHMM_DEVICE_EM", "0")
    monkeypatch.setenv("TEHMM_ESTEP_WIDE", "0")
    b = fresh()
    assert not b._can_fit_on_device(seqs)
    b.fit(seqs)
    assert_allclose(a._log_transmat, b._log_transmat, rtol=1e-6, atol=1e-9)
    assert_allclose(a.emissionModel.logProbs, b.emissionModel.logProbs, rtol=1e-6, atol=1e-9)
    assert_allclose(a._log_startprob, b._log_startprob, rtol=1e-6, atol=1e-9)


_DEVICE_ARRAYS_SCRIPT = r"""
import sys
import numpy as np
import torch
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel
model = synth.make_model(9, (3, 5, 4), (), seed=2)
lens = [300_000, 200_000]
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
total = int(offs[-1])
obs_h = synth.sample_obs(model, total, seed=4)
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
dev = torch.device("cuda", 0)
for rep in range(3):
    g = torch.Generator(device=dev)
    g.manual_seed(5 + rep)
    obs_d = torch.from_numpy(obs_h).to(dev)
    x = torch.rand(total, generator=g, device=dev, dtype=torch.float64)
    for _ in range(20):                                # keep the default stream busy right up to the create call
        x = torch.clamp(torch.sqrt(x * x), 0.0, 0.999)
    r_d = (torch.clamp(1 + torch.floor(torch.log1p(-x) / np.log(1 - 1 / 20.0)), max=100.0) / 20.0).contiguous()
    hb = HipBatch(obs_d.data_ptr(), offs, ratios=r_d.data_ptr(), device_ptrs=True, K=3)
    res = hm.eval(hb, viterbi=True, posterior=False, use_ratios=True)
    p_dev, lp_dev = hb.paths(), res["viterbi_logprob"].copy()
    hb.close()
    hb2 = HipBatch(obs_h, offs, r_d.cpu().numpy())
    res2 = hm.eval(hb2, viterbi=True, posterior=False, use_ratios=True)
    if not (np.isfinite(lp_dev).all() and np.array_equal(p_dev, hb2.paths()) and np.array_equal(lp_dev, res2["viterbi_logprob"])):
        print("MISMATCH in repetition", rep)
        sys.exit(1)
    hb2.close()
hm.close()
print("OK")
"""


def test_batch_from_device_arrays_waits_for_the_default_stream(tmp_path):
    """tehmm_batch_create with device pointers repacks on the batch's own non-blocking stream: arrays the caller has just
    queued work for on the default stream (torch tensors filled a moment ago) must be complete when they are read --
    the same decode as from host copies of the same arrays.  (Own process: torch's GPU context.)"""
    import os
    import subprocess
    import sys
    script = tmp_path / "device_arrays.py"
    script.write_text(_DEVICE_ARRAYS_SCRIPT)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_uint16_and_int32_tables_on_the_fused_path():
    """The reference's other two IntegerTrackTable types (_emission.pyx:82-144, 192-234): a uint16 / int32 table whose
    symbols fit a byte runs on the fused path (tehmm_batch_create_u16 / _i32) with the results of its uint8 copy, decode
    and E-step; a symbol beyond 255 is refused with TEHMM_ERR_UNSUPPORTED (MultitrackHmm then stays on the array-level
    entry points, which take all three types)."""
    from tehmm_amd import _lib, synth
    from tehmm_amd.engine import HipBatch, HipModel
    from tehmm_amd.hmm import MultitrackHmm
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    model = synth.make_model(7, (3, 5, 4), (), seed=8)
    lens = [30_000, 1, 12_345]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=9, missing=0.02)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    K, N, S = model.log_probs.shape
    ref = None
    for dt in (np.uint8, np.uint16, np.int32):
        hb = HipBatch(obs.astype(dt), offs)
        res = hm.eval(hb, viterbi=True, posterior=True)
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp = hm.estep(hb, False, start, trans, st)
        got = (hb.paths(), res["viterbi_logprob"].copy(), hb.posteriors(N), lp, trans, st)
        hb.close()
        if ref is None:
            ref = got
        else:
            for a, b in zip(got, ref):
                assert_array_equal(np.asarray(a), np.asarray(b))
    bad = obs.astype(np.uint16)
    bad[17, 1] = 300
    with pytest.raises(_lib.TeHmmHipError) as ei:
        HipBatch(bad, offs)
    assert ei.value.code == -3
    hm.close()
    em = IndependentMultinomialEmissionModel(7, [3, 5, 4])
    h = MultitrackHmm(em)
    assert h._can_fuse([obs.astype(np.int32)]) and not h._can_fuse([bad]) and not h._can_fuse([obs.astype(np.int64)])


@pytest.mark.timeout(900)
def test_config5_one_megabase_vs_oracle(monkeypatch):
    """BASELINE configs[4] at bench size: 100 states, 10 tracks, segment ratios, 1 Mb in 10 intervals of 100 kb, decode
    and score_samples in ONE evaluation (item-parallel posterior with the tile-layout emission rows taken from the log
    rows of the exact Viterbi, chunk-parallel exact Viterbi with both tie hypotheses): state paths and scores bit for
    bit, posteriors at 1e-6 (observed error asserted at 3e-7), forward log-likelihoods at 1e-9 against the oracle
    (oracle.eval_batch over all host threads: the reference's teHmmEval flow)."""
    import os
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in KNOBS + ("TEHMM_WIDE_VIT", "TEHMM_WIDE_CP", "TEHMM_WIDE_TOL"):
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(100, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
    lens = [100_000] * 10
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(offs[-1])
    obs = _noisy_obs(model, T, seed=33)
    ratios = np.ascontiguousarray(synth.random_ratios(T, seed=6))
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, viterbi=True, posterior=True, use_ratios=True)
    tm = hb.timing()
    paths, post = hb.paths(), hb.posteriors(100)
    hb.close()
    hm.close()
    assert tm.get("count:viterbi_chunk_jumps", 0) > 0 and "count:wide_chunk_parallel_warmup" in tm      # both chunk-parallel paths ran
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob, model.log_transmat, 1.0,
                                                  ratios, want_post=True, n_threads=max(1, min(16, os.cpu_count() or 1)))
    assert_array_equal(paths, p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    assert_allclose(res["forward_logprob"], flp_o, rtol=1e-9)
    worst = float(np.max(np.abs(post - post_o) / post_o))
    print("config 5, 1 Mb: posterior max rel err %.3g" % worst)
    assert_allclose(post, post_o, rtol=1e-6, atol=0)
    assert worst <= 3e-7
