"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C ABI,
against the CPU oracle and the reference's golden vectors.

Bars: Viterbi paths / log-probs and emission frames bit-exact; lattices, posteriors and
log-likelihoods within 1e-6 relative (BASELINE.json north_star), in practice far tighter."""
import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from conftest import golden_names, load_golden, ratios_of

pytestmark = pytest.mark.gpu

RTOL = 1e-6


@pytest.fixture(scope="module")
def hip():
    from tehmm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1, "no HIP device visible"
    return _lib


@pytest.mark.parametrize("name", golden_names("kern_"))
def test_array_level_vs_golden(hip, name):
    from tehmm_amd import _emission, _hmm
    from oracle import oracle
    g = load_golden(name)
    r = ratios_of(g)
    rows = g["rows"]
    T, K = g["obs"].shape
    N = g["lt"].shape[0]
    frame = np.zeros((T, N))
    _emission.fastAllLogProbs(g["obs"], g["log_probs"], frame, 1.0, r)
    assert_array_equal(frame[rows], g["frame_rows"])
    assert_array_equal(frame, oracle.emission(g["obs"], g["log_probs"], 1.0, r))
    frame_n = np.zeros((T, N))
    _emission.fastAllLogProbs(g["obs"], g["log_probs"], frame_n, 3.0 / K, r)
    assert_array_equal(frame_n[rows], g["frame_norm_rows"])
    fwd = np.zeros((T, N))
    _hmm._forward(T, N, g["pi"], g["lt"], frame, r, fwd)
    assert_allclose(fwd[rows], g["fwd_rows"], rtol=1e-9)
    bwd = np.zeros((T, N))
    _hmm._backward(T, N, g["pi"], g["lt"], frame, r, bwd)
    assert_allclose(bwd[rows], g["bwd_rows"], rtol=1e-9)
    path, lp = _hmm._viterbi(T, N, g["pi"], g["lt"], r, frame)
    assert path.dtype == np.int64
    assert_array_equal(path, g["vit_path"].astype(np.int64))
    assert lp == g["vit_logprob"]
    xi = np.zeros((N, N))
    _hmm._log_sum_lneta(T, N, fwd, g["lt"], bwd, frame, float(g["fwd_logprob"]), r, xi)
    assert_allclose(xi, g["xi_logsum"], rtol=1e-8, atol=1e-8)
    post_fit = oracle.posteriors(fwd, bwd, 0)
    stats = np.zeros_like(g["obs_stats"])
    _emission.fastAccumulateStats(g["obs"], stats, post_fit, r)
    assert_allclose(stats, g["obs_stats"], rtol=1e-8, atol=1e-300)


def test_emission_dtypes_and_quirk(hip):
    from tehmm_amd import _emission
    from oracle import oracle
    g = load_golden("quirk_q9_leading_rows")
    for dt in (np.uint8, np.uint16, np.int32):
        obs = g["obs"].astype(dt)
        out = np.zeros_like(g["frame"])
        _emission.fastAllLogProbs(obs, g["log_probs"], out, 1.0, None)
        assert_array_equal(out, g["frame"])
        assert_array_equal(out, oracle.emission(obs, g["log_probs"]))


def _eval(model_args, obs, offs, ratios=None, use_ratios=True):
    from tehmm_amd.engine import HipBatch, HipModel
    hm = HipModel(*model_args)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, viterbi=True, posterior=True, use_ratios=use_ratios)
    return res, hb.paths(), hb.posteriors(), hb


@pytest.mark.parametrize("name", golden_names("kern_"))
def test_fused_eval_vs_golden(hip, name):
    """decode + score_samples fused on the device vs the reference drivers' semantics:
    emission without ratios; Viterbi transitions with ratios; posteriors without."""
    from oracle import oracle
    g = load_golden(name)
    r = ratios_of(g)
    T = g["obs"].shape[0]
    offs = np.asarray([0, T], dtype=np.int64)
    res, paths, post, _ = _eval((g["lt"], g["pi"], g["log_probs"]), g["obs"], offs, r)
    lp_o, path_o = oracle.decode(g["obs"], g["log_probs"], g["pi"], g["lt"], 1.0, r)
    assert_array_equal(paths, path_o)
    assert res["viterbi_logprob"][0] == lp_o
    flp_o, post_o = oracle.score_samples(g["obs"], g["log_probs"], g["pi"], g["lt"])
    assert_allclose(res["forward_logprob"][0], flp_o, rtol=RTOL)
    assert_allclose(post, post_o, rtol=RTOL, atol=1e-15)
    if r is None:
        # same numbers as the reference itself produced
        assert_array_equal(paths, g["vit_path"].astype(np.int64))
        assert res["viterbi_logprob"][0] == g["vit_logprob"]
        assert_allclose(res["forward_logprob"][0], g["fwd_logprob"], rtol=RTOL)
        assert_allclose(post[g["rows"]], g["post_eval_rows"], rtol=RTOL, atol=1e-15)


def test_fused_known_answers(hip):
    g = load_golden("wikipedia")
    offs = np.asarray([0, 3], dtype=np.int64)
    res, paths, post, _ = _eval((g["lt"], g["pi"], g["log_probs"]), g["obs"], offs)
    assert_array_equal(paths, [1, 0, 0])                          # hmmTest.py:59
    assert abs(np.exp(res["viterbi_logprob"][0]) - 0.01344) < 1e-9    # hmmTest.py:58
    assert_allclose(post, g["post"], rtol=RTOL)
    res, paths, _, _ = _eval((g["lt"], g["pi"], g["log_probs4"]), g["obs4"], offs)
    assert_array_equal(paths, [1, 0, 0])
    assert res["viterbi_logprob"][0] == g["vit_logprob4"]


def test_fused_quirks(hip):
    g = load_golden("quirk_q1_zero_transitions")
    T = g["obs"].shape[0]
    offs = np.asarray([0, T], dtype=np.int64)
    res, paths, post, _ = _eval((g["lt"], g["pi"], g["log_probs"]), g["obs"], offs)
    assert_array_equal(paths, g["vit_path"])
    assert res["viterbi_logprob"][0] == g["vit_logprob"]
    assert_allclose(post, g["post"], rtol=RTOL, atol=1e-15)
    assert_allclose(res["forward_logprob"][0], g["fwd_logprob"], rtol=RTOL)
    g = load_golden("quirk_ties")
    T = g["obs"].shape[0]
    offs = np.asarray([0, T], dtype=np.int64)
    res, paths, _, _ = _eval((g["lt"], g["pi"], g["log_probs"]), g["obs"], offs)
    assert_array_equal(paths, g["vit_path"])
    assert res["viterbi_logprob"][0] == g["vit_logprob"]
    res, paths, _, _ = _eval((g["lt"], g["pi"], g["log_probs"]), g["obs"], offs, g["ratios"])
    assert_array_equal(paths, g["vit_path_r"])
    assert res["viterbi_logprob"][0] == g["vit_logprob_r"]
    g = load_golden("driver_asymmetry")
    T = g["obs"].shape[0]
    offs = np.asarray([0, T], dtype=np.int64)
    res, paths, post, _ = _eval((g["lt"], g["pi"], g["log_probs"]), g["obs"], offs, g["ratios"])
    assert_array_equal(paths, g["decode_path"])
    assert res["viterbi_logprob"][0] == g["decode_logprob"]
    assert_allclose(post, g["score_post"], rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("N,with_ratio", [(35, 0), (35, 1), (100, 0), (5, 1), (64, 0), (65, 1)])
def test_fused_batch_ragged(hip, N, with_ratio):
    """Many intervals of ragged lengths (including 1, 2, 3, 4, 5, 63..65, 255..258) in one batch."""
    from tehmm_amd import synth
    from oracle import oracle
    model = synth.make_model(N, seed=N + with_ratio, sparse=0.3 if N == 35 else 0.0)
    lens = [1, 2, 3, 4, 5, 17, 63, 64, 65, 255, 256, 257, 258, 1000, 1, 513, 31]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.03)
    ratios = synth.random_ratios(int(offs[-1]), seed=9) if with_ratio else None
    res, paths, post, hb = _eval((model.log_transmat, model.log_startprob, model.log_probs, 1.0,
                                  model.symbols_per_track), obs, offs, ratios)
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                  model.log_transmat, 1.0, ratios, n_threads=4)
    assert_array_equal(paths, p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
    assert_allclose(post, post_o, rtol=RTOL, atol=1e-15)
    assert_allclose(post.sum(axis=1), 1.0, rtol=1e-9)
    t = hb.timing()
    names = {k for k in t if not k.startswith("count:")}
    assert {"viterbi", "traceback"} <= names
    assert ({"forward", "backward_posterior"} <= names) or ({"forward_backward", "posterior_combine"} <= names) \
        or ({"forward_pass", "backward_posterior_pass", "backward_chain"} <= names)
    assert all(v >= 0.0 for v in t.values())


@pytest.mark.parametrize("N,with_ratio", [(35, 0), (35, 1), (6, 1), (20, 0)])
def test_fused_estep_vs_oracle(hip, N, with_ratio):
    """tehmm_estep_batch (basehmm.py:504-523 + hmm.py:545-574 fused on the device) against the
    oracle's per-sequence E-step, ragged lengths including 1, 2 and chunk-boundary sizes."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(N, seed=3 + N, sparse=0.2)
    lens = [1, 2, 700, 511, 512, 513, 1025, 64, 3]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=6, missing=0.03)
    ratios = synth.random_ratios(int(offs[-1]), seed=2) if with_ratio else None
    K, _, S = model.log_probs.shape
    start = np.zeros(N)
    trans = np.zeros((N, N))
    st = np.full((K, N, S), 0.5)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    lp = hm.estep(hb, bool(with_ratio), start, trans, st)
    seqs = [obs[offs[i]:offs[i + 1]] for i in range(len(lens))]
    rl = [ratios[offs[i]:offs[i + 1]] for i in range(len(lens))] if with_ratio else None
    ref = oracle.estep(seqs, model.log_probs, model.log_startprob, model.log_transmat, 1.0, rl)
    assert_allclose(lp, ref["logprob"], rtol=RTOL)
    assert_allclose(start, ref["start"], rtol=RTOL, atol=1e-13)
    assert_allclose(trans, ref["trans"], rtol=RTOL, atol=1e-13)
    assert_allclose(st, ref["obs"] + 0.5, rtol=RTOL, atol=1e-13)


def test_config2_single_long_interval(hip):
    """BASELINE config 2 shape: 35 states / 10 tracks, ONE long interval (2 Mb here so that the CPU
    oracle's Viterbi finishes in seconds).  Viterbi path + log-probability bit-exact against the
    oracle; posterior rows checked through size-independent properties (rows sum to 1, the
    posterior arg-max path has at least the oracle path's per-position posterior mass, the
    forward log-likelihood is reproduced by the oracle on a prefix)."""
    from tehmm_amd import synth
    from oracle import oracle
    model = synth.make_model(35, seed=0)
    T = 2_000_000
    rs = np.random.RandomState(5)
    # long observation column without the O(T) python sampler: tile a sampled 50 kb piece with
    # per-tile random symbol noise
    piece = synth.sample_obs(model, 50_000, seed=11)
    obs = np.tile(piece, (T // 50_000, 1))
    noise = rs.rand(T) < 0.2
    for k, sk in enumerate(model.symbols_per_track):
        col = obs[:, k]
        col[noise] = rs.randint(1, sk + 1, size=int(noise.sum()))
    offs = np.asarray([0, T], dtype=np.int64)
    res, paths, post, hb = _eval((model.log_transmat, model.log_startprob, model.log_probs, 1.0,
                                  model.symbols_per_track), obs, offs)
    lp_o, path_o = oracle.decode(obs, model.log_probs, model.log_startprob, model.log_transmat)
    assert_array_equal(paths, path_o)
    assert res["viterbi_logprob"][0] == lp_o
    assert_allclose(post.sum(axis=1), 1.0, rtol=1e-9)
    assert post.min() > 0 and np.isfinite(post).all()
    # prefix consistency of the forward log-likelihood (the chain is causal)
    Tp = 200_000
    flp_o, post_o = oracle.score_samples(obs[:Tp], model.log_probs, model.log_startprob,
                                         model.log_transmat)
    res_p, _, post_p, _ = _eval((model.log_transmat, model.log_startprob, model.log_probs, 1.0,
                                 model.symbols_per_track), obs[:Tp], np.asarray([0, Tp], dtype=np.int64))
    assert_allclose(res_p["forward_logprob"][0], flp_o, rtol=1e-9)
    assert_allclose(post_p, post_o, rtol=RTOL, atol=1e-15)
    # far from the prefix end the posteriors of the long run equal those of the prefix run
    assert_allclose(post[:Tp - 2000], post_p[:Tp - 2000], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("seed", range(12))
def test_fused_random_shapes(hip, seed):
    """Randomised shapes through the fused path: N from 1 to 63 (every padded tile size), 1-20
    tracks incl. 255-symbol tracks, emFac normalisation, sparse transitions (-1e100), missing data,
    ragged interval lengths, with and without segment ratios -- against the oracle drivers."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    rs = np.random.RandomState(1000 + seed)
    N = int(rs.choice([1, 2, 3, 4, 7, 8, 11, 15, 16, 17, 23, 31, 32, 33, 39, 40, 47, 55, 56, 63]))
    K = int(rs.randint(1, 21))
    syms = [int(rs.choice([1, 2, 3, 5, 17, 100, 255])) for _ in range(K)]
    gauss = [k for k in range(K) if syms[k] >= 100 and rs.rand() < 0.5]
    model = synth.make_model(N, syms, gauss, seed=seed, sparse=float(rs.choice([0.0, 0.3, 0.7])))
    normalize = float(rs.choice([1.0, 1.0, 3.0 / K]))
    n_iv = int(rs.randint(1, 9))
    lens = [int(x) for x in rs.choice([1, 2, 5, 63, 64, 65, 129, 300, 1000], size=n_iv)]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=seed, missing=0.05)
    with_ratio = bool(rs.rand() < 0.5)
    ratios = synth.random_ratios(int(offs[-1]), seed=seed) if with_ratio else None
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, normalize,
                  model.symbols_per_track if rs.rand() < 0.7 else None)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, viterbi=True, posterior=True)
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                  model.log_transmat, normalize, ratios, n_threads=4)
    assert_array_equal(hb.paths(), p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
    assert_allclose(hb.posteriors(), post_o, rtol=RTOL, atol=1e-15)
    # and the fused E-step on the same batch
    Kk, _, S = model.log_probs.shape
    start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((Kk, N, S))
    lp = hm.estep(hb, with_ratio, start, trans, st)
    seqs = [obs[offs[i]:offs[i + 1]] for i in range(n_iv)]
    rl = [ratios[offs[i]:offs[i + 1]] for i in range(n_iv)] if with_ratio else None
    ref = oracle.estep(seqs, model.log_probs, model.log_startprob, model.log_transmat, normalize, rl)
    assert_allclose(lp, ref["logprob"], rtol=RTOL)
    assert_allclose(start, ref["start"], rtol=RTOL, atol=1e-12)
    assert_allclose(trans, ref["trans"], rtol=RTOL, atol=1e-12)
    assert_allclose(st, ref["obs"], rtol=RTOL, atol=1e-12)


CHUNK_CONFIGS = [
    {"TEHMM_SPEC_CHUNK": "0"},                                                  # cooperative kernels only
    {"TEHMM_SPEC_CHUNK": "128", "TEHMM_LANE_SUB": "0"},                         # lane = state speculation
    {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64", "TEHMM_LANE_VIT": "0"},  # lane = item fwd/bwd + P0
    {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "128", "TEHMM_LANE_WARMUP": "24",
     "TEHMM_LANE_VIT": "0", "TEHMM_LANE_P0": "0"},                              # short warm-up: links fail
    {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"},                        # + lane = item exact Viterbi
    {"TEHMM_SPEC_CHUNK": "512", "TEHMM_LANE_SUB": "256", "TEHMM_LANE_MFMA": "1"},
    {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "128", "TEHMM_LANE_WARMUP": "24", "TEHMM_FB_RUNS": "0"},
    {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64", "TEHMM_VIT_RUNS": "0"},   # one verification per chunk
    {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64", "TEHMM_FUSED": "0"},      # round-1 posterior pipeline (K > 78)
]


@pytest.mark.parametrize("cfg", [1, 2, 4])
def test_chunk_parallel_sparse_model(hip, monkeypatch, cfg):
    """The chunk-parallel paths on a model with -1e100 (LOGZERO) transitions: states sitting at -1e100 are
    outside every binade, so the exact chains must refuse the speculative results wherever such a
    state is alive -- and the answers stay those of the reference."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0",
              "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in CHUNK_CONFIGS[cfg].items():
        monkeypatch.setenv(k, v)
    for sparse, seed in ((0.3, 5), (0.7, 6)):
        model = synth.make_model(35, seed=seed, sparse=sparse)
        lens = [300, 5000, 30000]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        obs = synth.sample_obs(model, int(offs[-1]), seed=seed, missing=0.03)
        hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=True, posterior=True)
        p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                      model.log_transmat, 1.0, None, n_threads=4)
        assert_array_equal(hb.paths(), p_o)
        assert_array_equal(res["viterbi_logprob"], vlp_o)
        assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
        assert_allclose(hb.posteriors(), post_o, rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("cfg", [1, 2, 4])
def test_chunk_parallel_deep_emission_drops(hip, monkeypatch, cfg):
    """Emission entries of -3e4 .. -2e5 (a symbol some states all but exclude): a state that falls that far
    behind inside one re-basing window stays inside its fp64 binade but no longer fits the exact
    integer representation of the quantised pass (index bits + 64 x offset < 2^53 u).  The speculative
    kernels must notice and leave such stretches to the exact chain."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0",
              "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in CHUNK_CONFIGS[cfg].items():
        monkeypatch.setenv(k, v)
    model = synth.make_model(12, seed=21)
    lp = model.log_probs.copy()
    lp[0, 0:4, 1] = -3.0e4          # track 0, symbol 1, states 0..3
    lp[1, 5:7, 2] = -2.0e5          # track 1, symbol 2, states 5, 6
    lens = [40000, 70000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=4, missing=0.02)
    hm = HipModel(model.log_transmat, model.log_startprob, lp, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    res = hm.eval(hb, viterbi=True, posterior=True)
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, lp, model.log_startprob, model.log_transmat,
                                                  1.0, None, n_threads=4)
    assert_array_equal(hb.paths(), p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
    assert_allclose(hb.posteriors(), post_o, rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("cfg", range(len(CHUNK_CONFIGS)))
@pytest.mark.parametrize("N", [35, 20, 7, 50, 60, 3])
def test_chunk_parallel_paths(hip, monkeypatch, cfg, N):
    """Every chunk-parallel code path (exact speculative Viterbi, speculative forward / backward,
    lane = item passes, their fix-up chains, jumps, ties, failed links) at chunk sizes small enough for
    the oracle: Viterbi paths and scores bit-exact, posteriors / log-likelihoods to 1e-6."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0",
              "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in CHUNK_CONFIGS[cfg].items():
        monkeypatch.setenv(k, v)
    model = synth.make_model(N, seed=3 + N)
    rs = np.random.RandomState(77 + cfg + N)
    lens = [int(x) for x in rs.choice([1, 63, 300, 1024, 2500, 4097, 6000, 9000], size=6)] + [20000, 45000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=cfg, missing=0.03)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    # N = 20 runs with segment ratios: decode applies them to the self-transitions (sequential exact
    # Viterbi kernel), the posterior ignores them (chunk-parallel passes) -- quirks Q11 / Q12
    ratios = synth.random_ratios(int(offs[-1]), seed=cfg) if N == 20 else None
    hb = HipBatch(obs, offs, ratios)
    for _ in range(2):                      # second call reuses the chunk / item workspaces
        res = hm.eval(hb, viterbi=True, posterior=True)
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                  model.log_transmat, 1.0, ratios, n_threads=4)
    assert_array_equal(hb.paths(), p_o)
    assert_array_equal(res["viterbi_logprob"], vlp_o)
    assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
    assert_allclose(hb.posteriors(), post_o, rtol=RTOL, atol=1e-15)
    t = hb.timing()
    if CHUNK_CONFIGS[cfg]["TEHMM_SPEC_CHUNK"] != "0" and ratios is None:
        # the long intervals reach |V| >= 2^18, so chunks really are jumped over (not just run exactly)
        assert "viterbi_speculate" in t and t["count:viterbi_exact_blocks"] > 0
        assert t["count:viterbi_chunk_jumps"] > 0
    if CHUNK_CONFIGS[cfg]["TEHMM_SPEC_CHUNK"] != "0":
        assert t["count:forward_chunk_jumps"] > 0 and t["count:backward_chunk_jumps"] > 0


def test_config_sizes_chunk_parallel_vs_sequential(hip, monkeypatch):
    """BASELINE-size check (config 2's single 10 Mb interval + config-3 style 0.2-2 Mb intervals, 35
    states x 10 tracks, ~24 Mb): the chunk-parallel default against the sequential cooperative kernels
    -- which the tests above hold to the oracle -- on the same batch.  Viterbi paths and scores must
    be bit-identical, forward log-likelihoods agree to 1e-9, posterior rows (sampled row ranges: interval
    starts and ends, chunk and item boundaries, interior) to 1e-6, every row sums to 1."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0",
              "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED"):
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
    rs = np.random.RandomState(123)
    lens = [10_000_000] + [int(x) for x in rs.randint(200_000, 2_000_000, size=13)]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offs[-1])
    piece = synth.sample_obs(model, 100_000, seed=3)
    obs = np.tile(piece, (total // 100_000 + 1, 1))[:total].copy()
    noise = rs.rand(total) < 0.25
    for k, sk in enumerate(model.symbols_per_track):
        col = obs[:, k]
        col[noise] = rs.randint(1, sk + 1, size=int(noise.sum()))
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)

    def run():
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=True, posterior=True)
        spans = []
        for i in range(len(lens)):
            a, b = int(offs[i]), int(offs[i + 1])
            spans += [(a, a + 700), (b - 700, b), (a + 4096 - 300, a + 4096 + 300), ((a + b) // 2, (a + b) // 2 + 600)]
        post = [hb.posteriors(35, r0, r1) for r0, r1 in spans]
        out = (hb.paths(), res["viterbi_logprob"].copy(), res["forward_logprob"].copy(), post, hb.timing())
        hb.close()
        return out

    p_a, v_a, f_a, post_a, t_a = run()
    assert t_a["count:viterbi_chunk_jumps"] > 1000 and t_a["count:forward_chunk_jumps"] >= len(lens)
    monkeypatch.setenv("TEHMM_SPEC_CHUNK", "0")
    p_b, v_b, f_b, post_b, t_b = run()
    assert "viterbi_speculate" not in t_b
    assert_array_equal(p_a, p_b)
    assert_array_equal(v_a, v_b)
    assert_allclose(f_a, f_b, rtol=1e-9)
    for x, y in zip(post_a, post_b):
        assert_allclose(x, y, rtol=RTOL, atol=1e-15)
        assert_allclose(x.sum(axis=1), 1.0, rtol=1e-9)


def test_chunk_parallel_call_sequences(hip, monkeypatch):
    """The same batch evaluated posterior-only, Viterbi-only, both, with changing chunk / item sizes in
    between (workspaces are built incrementally and rebuilt when the geometry changes)."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0",
              "TEHMM_LANE_MFMA", "TEHMM_FB_RUNS", "TEHMM_VIT_RUNS", "TEHMM_FUSED"):
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(35, seed=9)
    lens = [3000, 26000, 50000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=2, missing=0.03)
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob,
                                                  model.log_transmat, 1.0, None, n_threads=4)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    steps = [({"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"}, False, True),
             ({"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"}, True, False),
             ({"TEHMM_SPEC_CHUNK": "512", "TEHMM_LANE_SUB": "128"}, True, True),
             ({"TEHMM_SPEC_CHUNK": "512", "TEHMM_LANE_SUB": "128", "TEHMM_LANE_VIT": "0"}, True, True),
             ({"TEHMM_SPEC_CHUNK": "0"}, True, True),
             ({"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "256"}, True, True)]
    for env, vit, post in steps:
        for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_VIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res = hm.eval(hb, viterbi=vit, posterior=post)
        if vit:
            assert_array_equal(hb.paths(), p_o)
            assert_array_equal(res["viterbi_logprob"], vlp_o)
        if post:
            assert_allclose(res["forward_logprob"], flp_o, rtol=RTOL)
            assert_allclose(hb.posteriors(), post_o, rtol=RTOL, atol=1e-15)
