"""Round-3 GPU tests (through the C ABI): a fixed-seed slice of the chunk-parallel fuzz campaign, the observed
posterior error of the fused passes (f64 arithmetic, f32 alpha' storage) on a long sticky interval, the
pickle-free model format through eval and continued training, and the device-resident EM loop on two ranks."""
import os
import socket
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-6


@pytest.fixture(scope="module", autouse=True)
def hip():
    from tehmm_amd import _lib, build
    build.build()
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path has no CPU fallback")


# ------------------------------------------------------------------ fuzz slice (tools/fuzz_chunk_parallel.py)
def _fuzz():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_chunk_parallel
    return fuzz_chunk_parallel


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,seed0,n", [("plain", 5000, 28), ("bign", 7000, 12), ("long", 9000, 4)])
def test_fuzz_slice_vs_oracle(mode, seed0, n):
    """60 seeded cases of the campaign that found round 2's two chunk-parallel bugs: random models (2..63 states,
    with `bign` 64..128), track mixes, interval lengths, segment ratios, emFac and speculation knobs; paths and
    Viterbi scores bit-exact, forward log-likelihood and posteriors at 1e-6 against the CPU oracle."""
    fz = _fuzz()
    bad = [r for r in (fz.run_case(c, seed0, long_mode=(mode == "long"), bign=(mode == "bign"), verbose=False)
                       for c in range(n)) if r is not None]
    assert not bad, "\n".join(bad)


# ------------------------------------------------------------------ observed posterior error
def _tiled_obs(model, T, seed, piece_len=50_000, noise_p=0.2):
    from tehmm_amd import synth
    rs = np.random.RandomState(seed)
    piece = synth.sample_obs(model, piece_len, seed=seed + 1)
    obs = np.tile(piece, ((T + piece_len - 1) // piece_len, 1))[:T].copy()
    noise = rs.rand(T) < noise_p
    for k, sk in enumerate(model.symbols_per_track):
        obs[noise, k] = rs.randint(1, sk + 1, size=int(noise.sum()))
    return obs


@pytest.mark.timeout(900)
def test_posterior_error_on_long_sticky_interval():
    """One 1.2 Mb interval on the sticky model (self-transition 0.995), full posterior rows against the oracle.
    The fused passes compute in f64 and keep alpha' as f32: the observed maximum relative error is asserted
    (<= 2e-7, a tenth of the 1e-6 bar is not claimed) so that a regression of the margin shows."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(35, seed=4, stay=0.995)
    T = 1_200_000
    obs = _tiled_obs(model, T, seed=79)
    offs = np.asarray([0, T], dtype=np.int64)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    res = hm.eval(hb, viterbi=False, posterior=True)
    post = hb.posteriors()
    tm = hb.timing()
    hb.close()
    assert tm.get("count:backward_chunk_jumps", 1) > 0
    flp, post_o = oracle.score_samples(obs, model.log_probs, model.log_startprob, model.log_transmat)
    assert_allclose(res["forward_logprob"][0], flp, rtol=1e-9)
    rel = np.abs(post - post_o) / post_o
    worst = float(rel.max())
    print("posterior max rel error %.3g (mean %.3g) over %d x 35 cells" % (worst, float(rel.mean()), T))
    assert worst <= 2e-7


# ------------------------------------------------------------------ model format
def _gauss_hmm(seed=3):
    from tehmm_amd.emission import IndependentMultinomialAndGaussianEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    from tehmm_amd.track import CategoryMap, Track, TrackList
    gmap = CategoryMap(reserved=1, defaultVal="0", scale=0.5)
    for v in range(0, 60, 2):
        gmap.getMap(v, update=True)
    gmap.sort()
    tracks = TrackList([Track("cat", 0), Track("gauss", 1, dist="gaussian", valueMap=gmap), Track("cat2", 2)])
    em = IndependentMultinomialAndGaussianEmissionModel(5, [4, 30, 6], tracks,
                                                        random_state=np.random.RandomState(seed), randomize=True)
    h = MultitrackHmm(em, n_iter=3, thresh=0.0, fixStart=False)
    h.trackList = tracks
    rs = np.random.RandomState(seed + 1)
    tm = rs.rand(5, 5) + 3 * np.eye(5)
    h.transmat_ = tm / tm.sum(axis=1, keepdims=True)
    h.init_params = ""
    return h


def test_model_io_eval_and_continued_fit(tmp_path):
    """save -> load -> (a) identical evaluation, (b) continued training takes the same steps as the model that
    was never saved, gaussian refit included (a reloaded gaussian model used to continue as a multinomial)."""
    from tehmm_amd import modelIO
    rs = np.random.RandomState(9)
    seqs = [np.stack([rs.randint(1, 5, size=T), rs.randint(1, 31, size=T), rs.randint(1, 7, size=T)],
                     axis=1).astype(np.uint8) for T in (900, 2500, 400)]
    a = _gauss_hmm()
    path = str(tmp_path / "m.npz")
    modelIO.saveModel(path, a)
    b = modelIO.loadModel(path)
    ra = a._eval_tables(seqs, True, True)
    rb = b._eval_tables(seqs, True, True)
    for x, y in zip(ra["paths"], rb["paths"]):
        assert_array_equal(x, y)
    assert_array_equal(ra["viterbi_logprob"], rb["viterbi_logprob"])
    for x, y in zip(ra["posteriors"], rb["posteriors"]):
        assert_array_equal(x, y)
    assert b._can_fit_on_device(seqs)
    a.fit(seqs)
    b.init_params = ""
    b.fit(seqs)
    assert_allclose(b._log_transmat, a._log_transmat, rtol=1e-12)
    assert_allclose(b.emissionModel.logProbs, a.emissionModel.logProbs, rtol=1e-12)
    assert_allclose(b.emissionModel.gaussParams, a.emissionModel.gaussParams, rtol=1e-12)
    assert np.abs(b.emissionModel.gaussParams[1]).max() > 0
    # fixEmission on a gaussian model: the (mu, sigma) table must survive the M-steps
    c = modelIO.loadModel(path)
    c.fixEmission = True
    c.init_params = ""
    gp0 = c.emissionModel.gaussParams.copy()
    c.fit(seqs)
    assert_array_equal(c.emissionModel.gaussParams, gp0)


# ------------------------------------------------------------------ device-resident EM on two ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _em_setup(n_tables=7):
    from tehmm_amd import synth
    from tehmm_amd.emission import IndependentMultinomialEmissionModel
    from tehmm_amd.hmm import MultitrackHmm
    model = synth.make_model(6, (3, 5, 4), (), seed=21)
    start = synth.make_model(6, (3, 5, 4), (), seed=22)
    lens = [400, 150, 3000, 77, 2200, 64, 900][:n_tables]
    seqs = [synth.sample_obs(model, T, seed=30 + i, missing=0.03) for i, T in enumerate(lens)]
    em = IndependentMultinomialEmissionModel(6, [3, 5, 4])
    em.logProbs = start.log_probs.copy()
    h = MultitrackHmm(em, n_iter=4, thresh=0.0, maxProb=True, fixStart=False)
    h.transmat_ = start.transmat.copy()
    h.init_params = ""
    return h, seqs


def _em_worker(rank, world, port, n_tables, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tehmm_amd.track import TrackData
        h, seqs = _em_setup(n_tables)
        assert h._can_fit_on_device(seqs)
        h.train(TrackData(seqs, None, [3, 5, 4]))             # every rank passes the SAME list; fit shards it
        q.put((rank, h._log_transmat, h.emissionModel.logProbs, h._log_startprob, h.best_forward_log_prob,
               h.last_forward_log_prob, h.bestCopy.current_iteration, h.current_iteration))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_tables", [7, 1])
def test_device_em_two_ranks_match_single_process(n_tables):
    """MultitrackHmm.fit on two ranks sharing the box's GPU (gloo: the statistics buffer is staged through the
    host): the tables are LPT-sharded inside _fit_device, the ranks meet in one all-reduce per iteration, the
    --maxProb bookkeeping sees the all-gathered per-sequence log-likelihoods -- both ranks end with the same
    parameters and the same best iteration as a single process, at 1e-6.  n_tables = 1: rank 1's shard is
    EMPTY and it still takes part in every collective."""
    import torch.multiprocessing as mp
    from tehmm_amd.track import TrackData
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_em_worker, args=(r, 2, port, n_tables, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(2)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    h, seqs = _em_setup(n_tables)
    h.train(TrackData(seqs, None, [3, 5, 4]))
    for rank, lt, lp, pi, best, last, best_it, it in got:
        assert_allclose(lt, h._log_transmat, rtol=RTOL)
        assert_allclose(lp, h.emissionModel.logProbs, rtol=RTOL, atol=1e-9)
        assert_allclose(pi, h._log_startprob, rtol=RTOL)
        assert_allclose(best, h.best_forward_log_prob, rtol=1e-9)
        assert_allclose(last, h.last_forward_log_prob, rtol=1e-9)
        assert best_it == h.bestCopy.current_iteration and it == h.current_iteration
    assert_array_equal(got[0][1], got[1][1])                  # the two ranks agree bit for bit


# ------------------------------------------------------------------ chunk-parallel (fused) E-step
def _rel(a, b, floor=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.abs(b) > floor
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,symbols,gauss,normalize,small", [
    (35, None, None, 1.0, None),                          # BASELINE configs[3] model: 10 multinomial + 2 gaussian
    (35, None, None, 1.0, "0"),                           # every track through the LDS histograms
    (35, None, None, 1.0, "300"),                         # every track through the one-hot matrix product
    (5, (3, 5, 4), (), 1.0, None),
    (20, (2, 17, 255, 100, 3), (2,), 1.0, None),
    (50, (4, 250, 30, 250, 250), (1, 3, 4), 1.0, None),   # three 250-bin tracks at 52 padded states: two LDS groups
    (63, (2, 2, 3, 250), (3,), 1.0, None),
    (35, (3, 5, 4, 30), (), 0.5, None),                   # --emFac: log-domain emission rows in the passes
])
def test_fused_estep_vs_oracle(monkeypatch, N, symbols, gauss, normalize, small):
    """tehmm_estep_batch on the chunk-parallel path (k_fused_fwd, k_fused_bwd<ESTEP>, the exact chains, k_estep_xi /
    k_estep_hist_mfma / k_estep_hist_lds) against the oracle's per-sequence E-step (basehmm.py:504-523,
    hmm.py:545-574) on ragged intervals -- first / last chunks, tails shorter than a chunk, one-row and sub-chunk
    intervals walked by the exact chains -- and against the sequential path.  Statistics at 1e-6; the observed
    error (floats rows, fp64 sums) is asserted at 3e-7 so that a loss of margin shows."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_ESTEP_FUSED", "TEHMM_ESTEP_SMALL"):
        monkeypatch.delenv(k, raising=False)
    if small is not None:
        monkeypatch.setenv("TEHMM_ESTEP_SMALL", small)
    if symbols is None:
        model = synth.make_model(N, synth.CONFIG4_SYMBOLS, synth.CONFIG4_GAUSSIAN, seed=12)
    else:
        model = synth.make_model(N, symbols, gauss, seed=12 + N)
    rs = np.random.RandomState(3 + N)
    lens = [int(x) for x in rs.randint(20000, 45000, size=5)] + [1, 70, 1500, 1024, 2048 + 64, 4097]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
    K, _, S = model.log_probs.shape
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, normalize, model.symbols_per_track)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TEHMM_ESTEP_FUSED", mode)
        hb = HipBatch(obs, offs)
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp = hm.estep(hb, False, start, trans, st)
        tm = hb.timing()
        got[mode] = (lp, start, trans, st, hb.interval_logprobs())
        hb.close()
        assert ("estep_reduce" in tm) == (mode == "1")
        if mode == "1":
            assert tm["count:backward_chunk_jumps"] > 0 and tm["count:forward_chunk_jumps"] > 0
    ref = oracle.estep([obs[offs[i]:offs[i + 1]] for i in range(len(lens))], model.log_probs, model.log_startprob,
                       model.log_transmat, normalize, None)
    lp, start, trans, st, ilp = got["1"]
    assert_allclose(lp, ref["logprob"], rtol=1e-9)
    assert_allclose(ilp.sum(), ref["logprob"], rtol=1e-9)
    assert_allclose(start, ref["start"], rtol=RTOL, atol=1e-12)
    assert_allclose(trans, ref["trans"], rtol=RTOL, atol=1e-9)
    assert_allclose(st, ref["obs"], rtol=RTOL, atol=1e-9)
    worst = max(_rel(start, ref["start"]), _rel(trans, ref["trans"], 1e-6), _rel(st, ref["obs"], 1e-6))
    print("fused E-step N=%d: max rel error of the statistics %.3g" % (N, worst))
    assert worst <= 3e-7
    assert_allclose(got["1"][4], got["0"][4], rtol=1e-9)          # per-interval log-likelihoods of the two paths
    assert_allclose(got["0"][2], ref["trans"], rtol=RTOL, atol=1e-9)


def test_fused_estep_impossible_rows_poison_like_the_reference(monkeypatch):
    """An interval with a row no state can emit AFTER its first emittable one: the reference's lattices are NaN from
    there on, so are its statistics.  The fused path must say so too (the forward chain flags the interval)."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    monkeypatch.delenv("TEHMM_ESTEP_FUSED", raising=False)
    model = synth.make_model(6, (3, 5, 4), (), seed=21)
    lp3 = model.log_probs.copy()
    lp3[1, :, 2] = -np.inf                      # symbol 2 of track 1 cannot be emitted by any state
    lens = [5000, 3000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=6)
    obs[obs[:, 1] == 2, 1] = 1
    obs[6500, 1] = 2                            # inside the second interval
    hm = HipModel(model.log_transmat, model.log_startprob, lp3, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs)
    K, N, S = lp3.shape
    start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
    lp = hm.estep(hb, False, start, trans, st)
    ilp = hb.interval_logprobs()
    assert "estep_reduce" in hb.timing()
    hb.close()
    assert np.isnan(lp) and np.isnan(trans).all() and np.isnan(st).any()
    assert np.isfinite(ilp[0]) and np.isnan(ilp[1])


# ------------------------------------------------------------------ segment ratios on the lane path, 37 <= N <= 63
@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,env", [
    (41, {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"}),
    (50, {"TEHMM_SPEC_CHUNK": "512", "TEHMM_LANE_SUB": "128"}),
    (63, {"TEHMM_SPEC_CHUNK": "256", "TEHMM_LANE_SUB": "64"}),
    (63, {"TEHMM_SPEC_CHUNK": "1024"}),
])
def test_ratio_decode_37_to_63_states_chunk_parallel(monkeypatch, N, env):
    """Decode on a segmented table (_hmm.pyx:229-247, from-state-0 quirk Q4) for 37..63 states: round 2 sent these to
    the sequential kernel; the quantised lane pass, its P0 and the exact chain now exist with ratios up to 64 padded
    states.  Paths and scores bit-exact against the oracle, and chunks really are jumped over."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_SPEC_CHUNK", "TEHMM_LANE_SUB", "TEHMM_LANE_WARMUP", "TEHMM_LANE_VIT", "TEHMM_LANE_P0", "TEHMM_VIT_RUNS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model = synth.make_model(N, seed=11 + N)
    lens = [1, 300, 5000, 60000, 150000] if env["TEHMM_SPEC_CHUNK"] != "1024" else [700, 90000, 300000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(offs[-1])
    obs = synth.sample_obs(model, T, seed=N, missing=0.02)
    rs = np.random.RandomState(N)
    ratios = synth.random_ratios(T, seed=N)
    ratios[rs.rand(T) < 0.4] = 1.0
    idx = rs.randint(0, T, size=max(1, T // 500))
    ratios[idx] = rs.randint(200, 5000, size=idx.size).astype(np.float64) / 20.0
    ratios = np.ascontiguousarray(ratios)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, viterbi=True, posterior=False, use_ratios=True)
    paths = hb.paths()
    tm = hb.timing()
    hb.close()
    assert tm.get("count:viterbi_chunk_jumps", 0) > 0
    for i in range(len(lens)):
        a, b = int(offs[i]), int(offs[i + 1])
        lp_o, path_o = oracle.decode(obs[a:b], model.log_probs, model.log_startprob, model.log_transmat, 1.0, ratios[a:b])
        assert_array_equal(paths[a:b], path_o)
        assert res["viterbi_logprob"][i] == lp_o


# ------------------------------------------------------------------ chunk-parallel posterior, 64 <= N <= 128
@pytest.mark.timeout(900)
@pytest.mark.parametrize("N,kw,normalize", [
    (64, {}, 1.0), (77, {}, 1.0), (100, {}, 1.0), (128, {}, 1.0),
    (100, dict(stay=0.995), 1.0),               # sticky: the warm-up has to grow until every link verifies
    (100, dict(sparse=0.5), 1.0),               # -1e100 transitions
    (90, {}, 0.3),                              # --emFac
])
def test_wide_posterior_chunk_parallel_vs_oracle(monkeypatch, N, kw, normalize):
    """BaseHMM.score_samples for the model sizes of BASELINE configs[4] (64..128 states) on the item-parallel
    matrix-core passes of tehmm_wide.hip.h: posteriors and forward log-likelihoods against the CPU oracle on
    ragged intervals (one-row, shorter than an item, not a multiple of the item length), 1e-6 -- and the
    same batch on the sequential kernels (TEHMM_WIDE_CP=0)."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_WIDE_CP", "TEHMM_WIDE_SUB", "TEHMM_LANE_WARMUP"):
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(N, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=3 + N, **kw)
    rs = np.random.RandomState(N)
    lens = [int(x) for x in rs.randint(9000, 30000, size=4)] + [1, 70, 1500, 64, 129]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, normalize, model.symbols_per_track)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TEHMM_WIDE_CP", mode)
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=False, posterior=True)
        tm = hb.timing()
        got[mode] = (res["forward_logprob"].copy(), np.array(hb.posteriors()))
        hb.close()
        assert ("count:wide_chunk_parallel_warmup" in tm) == (mode == "1")
    _, _, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob, model.log_transmat,
                                            normalize, None, n_threads=8)
    for mode in ("1", "0"):
        flp, post = got[mode]
        assert_allclose(flp, flp_o, rtol=1e-9)
        assert_allclose(post, post_o, rtol=RTOL, atol=1e-15)
    worst = float(np.max(np.abs(got["1"][1] - post_o) / post_o))
    print("wide posterior N=%d: max rel error %.3g" % (N, worst))
    assert worst <= 3e-7


def test_wide_posterior_impossible_rows_fall_back(monkeypatch):
    """Rows no state can emit (leading ones: quirk Q9; later ones: NaN lattices) are the sequential kernels' business:
    the chunk-parallel path must notice them and step aside, results equal to the sequential run bit for bit."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    monkeypatch.delenv("TEHMM_WIDE_CP", raising=False)
    model = synth.make_model(70, (3, 5, 4), (), seed=21)
    lp3 = model.log_probs.copy()
    lp3[1, :, 2] = -np.inf
    lens = [6000, 5000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=6)
    obs[obs[:, 1] == 2, 1] = 1
    obs[0:3, 1] = 2                             # leading impossible rows of the first interval (zeroed, Q9)
    obs[8000, 1] = 2                            # inside the second interval: NaN from there on
    hm = HipModel(model.log_transmat, model.log_startprob, lp3, 1.0, model.symbols_per_track)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TEHMM_WIDE_CP", mode)
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=False, posterior=True)
        assert "count:wide_chunk_parallel_warmup" not in hb.timing()
        out[mode] = (res["forward_logprob"].copy(), np.array(hb.posteriors()))
        hb.close()
    assert_array_equal(out["1"][0], out["0"][0])
    assert_array_equal(out["1"][1], out["0"][1])
    assert np.isfinite(out["1"][0][0])


# ------------------------------------------------------------------ chunk-parallel exact Viterbi, 64 <= N <= 128
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("N,with_ratio,kw", [
    (100, True, {}),                        # BASELINE configs[4]: 100 states, segment ratios on both sides of 1
    (100, False, {}),
    (64, True, {}), (77, False, {}), (128, True, {}),
    (100, True, dict(stay=0.995)),          # sticky
    (90, True, dict(sparse=0.5)),           # -1e100 transitions
])
def test_wide_viterbi_chunk_parallel_bit_exact(monkeypatch, N, with_ratio, kw):
    """decode (basehmm.py:301-330 over _hmm._viterbi, _hmm.pyx:201-259, incl. the from-state-0 ratio quirk Q4) for
    64..128 states on the chunk-parallel path of tehmm_wide.hip.h -- plain P0, quantised P2 carrying both rounding-tie
    hypotheses, exact four-wave chain with verified jumps: state paths and scores bit for bit against the CPU oracle,
    chunks really jumped over, and the same on the sequential kernel."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_WIDE_VIT", "TEHMM_WIDE_CP", "TEHMM_SPEC_CHUNK"):
        monkeypatch.delenv(k, raising=False)
    model = synth.make_model(N, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=5 + N, **kw)
    rs = np.random.RandomState(N)
    lens = [int(x) for x in rs.randint(30000, 90000, size=3)] + [1, 70, 1500, 2048, 5000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(offs[-1])
    obs = synth.sample_obs(model, T, seed=7, missing=0.02)
    ratios = None
    if with_ratio:
        ratios = synth.random_ratios(T, seed=N)
        ratios[rs.rand(T) < 0.4] = 1.0
        idx = rs.randint(0, T, size=max(1, T // 500))
        ratios[idx] = rs.randint(200, 5000, size=idx.size).astype(np.float64) / 20.0
        ratios = np.ascontiguousarray(ratios)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TEHMM_WIDE_VIT", mode)
        hb = HipBatch(obs, offs, ratios)
        res = hm.eval(hb, viterbi=True, posterior=False, use_ratios=with_ratio)
        tm = hb.timing()
        got[mode] = (res["viterbi_logprob"].copy(), np.array(hb.paths()))
        hb.close()
        if mode == "1":
            assert tm.get("count:viterbi_chunk_jumps", 0) > 0
            print("wide Viterbi N=%d ratio=%d: %d exact blocks of 16, %d chunk jumps, %d positions"
                  % (N, with_ratio, tm["count:viterbi_exact_blocks"], tm["count:viterbi_chunk_jumps"], T))
        else:
            assert "count:viterbi_chunk_jumps" not in tm
    for i in range(len(lens)):
        a, b = int(offs[i]), int(offs[i + 1])
        lp_o, path_o = oracle.decode(obs[a:b], model.log_probs, model.log_startprob, model.log_transmat, 1.0,
                                     None if ratios is None else ratios[a:b])
        for mode in ("1", "0"):
            assert_array_equal(got[mode][1][a:b], path_o)
            assert got[mode][0][i] == lp_o


@pytest.mark.timeout(900)
def test_wide_viterbi_chain_forms_agree(monkeypatch):
    """The exact chain of the 64..128-state Viterbi in its three forms -- following the quantised pass chunk by chunk
    (ready flags, the default for few intervals), behind the pass with the interval heads walked beside it
    (TEHMM_WIDE_POLL=0), and in one launch (TEHMM_WIDE_HEAD=0) -- and both results requested at once (the posterior
    passes then share the GPU with the chain): identical paths and scores, equal to the oracle's; posteriors of the
    combined evaluation equal to those of a posterior-only one."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    for k in ("TEHMM_WIDE_VIT", "TEHMM_WIDE_CP", "TEHMM_SPEC_CHUNK", "TEHMM_WIDE_POLL", "TEHMM_WIDE_HEAD", "TEHMM_DEFER"):
        monkeypatch.delenv(k, raising=False)
    N = 100
    model = synth.make_model(N, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=11)
    rs = np.random.RandomState(4)
    lens = [52000, 38000, 47000, 700, 9000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(offs[-1])
    obs = synth.sample_obs(model, T, seed=9, missing=0.02)
    ratios = synth.random_ratios(T, seed=2)
    ratios[rs.rand(T) < 0.4] = 1.0
    ratios = np.ascontiguousarray(ratios)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    got = {}
    for tag, env in (("follow", {}), ("behind", {"TEHMM_WIDE_POLL": "0"}), ("one_launch", {"TEHMM_WIDE_HEAD": "0"})):
        for k in ("TEHMM_WIDE_POLL", "TEHMM_WIDE_HEAD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hb = HipBatch(obs, offs, ratios)
        res = hm.eval(hb, viterbi=True, posterior=False, use_ratios=True)
        tm = hb.timing()
        assert tm.get("count:viterbi_chunk_jumps", 0) > 0
        got[tag] = (res["viterbi_logprob"].copy(), np.array(hb.paths()))
        hb.close()
    for k in ("TEHMM_WIDE_POLL", "TEHMM_WIDE_HEAD"):
        monkeypatch.delenv(k, raising=False)
    hb = HipBatch(obs, offs, ratios)
    res = hm.eval(hb, viterbi=True, posterior=True, use_ratios=True)
    tm = hb.timing()
    assert tm.get("count:viterbi_chunk_jumps", 0) > 0 and tm.get("count:wide_chunk_parallel_warmup", 0) > 0
    got["both"] = (res["viterbi_logprob"].copy(), np.array(hb.paths()))
    post_both, flp_both = np.array(hb.posteriors()), res["forward_logprob"].copy()
    res = hm.eval(hb, viterbi=False, posterior=True, use_ratios=True)
    assert_array_equal(np.array(hb.posteriors()), post_both)
    assert_array_equal(res["forward_logprob"], flp_both)
    hb.close()
    for tag in ("behind", "one_launch", "both"):
        assert_array_equal(got[tag][1], got["follow"][1])
        assert_array_equal(got[tag][0], got["follow"][0])
    for i in (1, 3):
        a, b = int(offs[i]), int(offs[i + 1])
        lp_o, path_o = oracle.decode(obs[a:b], model.log_probs, model.log_startprob, model.log_transmat, 1.0, ratios[a:b])
        assert_array_equal(got["follow"][1][a:b], path_o)
        assert got["follow"][0][i] == lp_o


@pytest.mark.timeout(600)
def test_posterior_viterbi_order_modes_agree(monkeypatch):
    """TEHMM_DEFER only orders the two pipelines of tehmm_eval_batch on the GPU (posterior passes behind the Viterbi
    passes, behind the emission rows, split around the quantised pass, or unordered): paths, scores, posteriors and
    log-likelihoods are identical bit for bit in every mode, and equal to the oracle's on one interval."""
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=3)
    lens = [150000, 90000, 1, 4000, 260000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=21, missing=0.02)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    got = {}
    for mode in ("1", "0", "2", "3"):
        monkeypatch.setenv("TEHMM_DEFER", mode)
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=True, posterior=True)
        tm = hb.timing()
        assert tm.get("count:viterbi_chunk_jumps", 0) > 0 and tm.get("count:forward_chunk_jumps", 0) > 0
        got[mode] = (res["viterbi_logprob"].copy(), res["forward_logprob"].copy(), np.array(hb.paths()),
                     np.array(hb.posteriors()))
        hb.close()
    monkeypatch.delenv("TEHMM_DEFER")
    for mode in ("0", "2", "3"):
        for a, b in zip(got[mode], got["1"]):
            assert_array_equal(a, b)
    a, b = int(offs[1]), int(offs[2])
    lp_o, path_o = oracle.decode(obs[a:b], model.log_probs, model.log_startprob, model.log_transmat, 1.0, None)
    assert_array_equal(got["3"][2][a:b], path_o)
    assert got["3"][0][1] == lp_o


@pytest.mark.timeout(600)
def test_device_block_pool_reuse_and_trim():
    """Workspaces of a destroyed batch are handed to the next one (DevPool, tehmm_trim_pools): a second fresh batch of
    the same shape -- on blocks full of the first one's data -- and a third one after the pools were emptied give
    the first batch's results bit for bit; a batch of another model in between does not disturb them."""
    from tehmm_amd import _lib, synth
    from tehmm_amd.engine import HipBatch, HipModel
    model = synth.make_model(35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=8)
    other = synth.make_model(20, (3, 5, 7), (), seed=9)
    lens = [70000, 1, 123456, 300]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=4, missing=0.02)
    obs_o = synth.sample_obs(other, 50000, seed=5)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    ho = HipModel(other.log_transmat, other.log_startprob, other.log_probs, 1.0, other.symbols_per_track)

    def run():
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=True, posterior=True)
        out = (res["viterbi_logprob"].copy(), res["forward_logprob"].copy(), np.array(hb.paths()), np.array(hb.posteriors()))
        hb.close()
        return out

    first = run()
    hbo = HipBatch(obs_o, np.asarray([0, 50000], dtype=np.int64))
    ho.eval(hbo, viterbi=True, posterior=True)
    hbo.close()
    second = run()
    _lib.trim_pools()
    third = run()
    for a, b, c in zip(first, second, third):
        assert_array_equal(a, b)
        assert_array_equal(a, c)
