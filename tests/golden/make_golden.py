#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REAL REFERENCE (glennhickey/teHmm) in this container.

Only runnable where /root/reference exists (the build container).  It
  1. copies the reference's hot-path sources into a fresh temp directory OUTSIDE the repo,
  2. applies the mechanical Python-2 -> 3 / Cython-3 patches SURVEY.md section 8c lists,
  3. cythonizes _hmm / _basehmm / _emission / _track, imports the package, and
  4. runs the reference functions on seeded inputs, writing inputs + outputs as .npz here.

Nothing of the reference (source, bytecode, binaries) is written into the repository:
only the numeric inputs/outputs below.  Re-run:  python tests/golden/make_golden.py
"""
import collections
import collections.abc
import glob
import hashlib
import importlib
import os
import random
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TEHMM_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

PY_FILES = ["__init__.py", "basehmm.py", "hmm.py", "emission.py", "track.py", "trackIO.py",
            "common.py", "modelIO.py"]
PYX_FILES = ["_hmm.pyx", "_basehmm.pyx", "_emission.pyx", "_track.pyx"]

SETUP_PY = r"""
import numpy
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = [Extension("teHmm." + n, ["teHmm/" + n + ".pyx"], include_dirs=[numpy.get_include()])
        for n in ("_hmm", "_basehmm", "_emission", "_track")]
setup(name="teHmm_ref_scratch", ext_modules=cythonize(exts, language_level=2, quiet=True))
"""


def build_reference():
    root = tempfile.mkdtemp(prefix="tehmm_ref_")
    pkg = os.path.join(root, "teHmm")
    os.makedirs(pkg)
    for f in PY_FILES + PYX_FILES:
        shutil.copy(os.path.join(REF, f), os.path.join(pkg, f))
    # Cython 3 dropped np.int_t / np.int
    for f in ("_hmm.pyx", "_basehmm.pyx"):
        p = os.path.join(pkg, f)
        s = open(p).read().replace("np.int_t", "np.int64_t").replace("dtype=np.int)", "dtype=np.int64)")
        open(p, "w").write(s)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", pkg],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = os.path.join(pkg, "track.py")
    s = open(p).read().replace("from _track import runSum", "from ._track import runSum")
    open(p, "w").write(s)
    # pybedtools stub (BED IO is never exercised here)
    stub = os.path.join(root, "pybedtools")
    os.makedirs(stub)
    open(os.path.join(stub, "__init__.py"), "w").write(
        "__version__='0.6.9'\nclass BedTool(object):\n    pass\nclass Interval(object):\n    pass\n"
        "def set_tempdir(*a, **k):\n    pass\ndef cleanup(*a, **k):\n    pass\n")
    open(os.path.join(root, "setup.py"), "w").write(SETUP_PY)
    subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=root,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # runtime shims for NumPy 2 / Python 3.10
    np.float = np.float64
    np.int = np.int64
    np.alltrue = np.all
    collections.Iterable = collections.abc.Iterable
    sys.path.insert(0, root)
    return root


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-40s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def rows_subset(T):
    return np.unique(np.asarray([0, 1, 2, T // 3, T // 2, T - 3, T - 2, T - 1]).clip(0, T - 1))


def main():
    root = build_reference()
    from teHmm import _hmm, _basehmm, _emission            # noqa: the compiled reference kernels
    from teHmm.hmm import MultitrackHmm
    from teHmm.basehmm import MultinomialHMM, BaseHMM, logsumexp
    from teHmm.emission import IndependentMultinomialEmissionModel
    from teHmm.track import IntegerTrackTable
    from teHmm.common import myLog
    from tehmm_amd import synth

    def make_ref_hmm(model, eff_len=None, **kw):
        em = IndependentMultinomialEmissionModel(model.n_states, model.symbols_per_track,
                                                 effectiveSegmentLength=eff_len)
        em.logProbs = model.log_probs.copy()
        h = MultitrackHmm(em, **kw)
        h.transmat_ = model.transmat.copy()       # property -> myLog (hmm.py:625-645)
        h.startprob_ = np.exp(model.log_startprob)
        return h, em

    def segmented_table(obs, lens):
        """IntegerTrackTable with segOffsets so getSegmentLengthsAsRatio works (track.py:504-513)."""
        T, K = obs.shape
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        tab = IntegerTrackTable(K, "chrS", 0, int(np.sum(lens)))
        tab.data = obs.copy()
        tab.segOffsets = offs
        tab.shape = (T, K)
        return tab

    # ------------------------------------------------------------------ 1. Wikipedia HMM
    emissionprob = [[0.1, 0.4, 0.5], [0.6, 0.3, 0.1]]
    startprob = [0.6, 0.4]
    transmat = [[0.7, 0.3], [0.4, 0.6]]
    h = MultinomialHMM(2, startprob=startprob, transmat=transmat)
    h.emissionprob_ = emissionprob
    lp_tw, path_tw = h.decode([0, 1, 2])
    post_tw = h.predict_proba([0, 1, 2])
    em = IndependentMultinomialEmissionModel(2, [3], [emissionprob], zeroAsMissingData=False)
    th = MultitrackHmm(em, startprob=startprob, transmat=transmat)
    obs = np.asarray([[0], [1], [2]], dtype=np.uint8)
    frame = th._compute_log_likelihood(obs)
    lp, path = th.decode(obs)
    flp, post = th.score_samples(obs)
    _, fwd = th._do_forward_pass(frame)
    bwd = th._do_backward_pass(frame)
    # 4-track variant (hmmTest.py:104-135)
    obs4 = np.asarray([[0, 0, 0, 0], [1, 0, 0, 5], [2, 0, 0, 7]], dtype=np.uint8)
    ep4 = [emissionprob, [[1.], [1.]], [[1.], [1.]], [[.1] * 10, [.1] * 10]]
    em4 = IndependentMultinomialEmissionModel(2, [3, 1, 1, 10], ep4, zeroAsMissingData=False)
    th4 = MultitrackHmm(em4, startprob=startprob, transmat=transmat)
    lp4, path4 = th4.decode(obs4)
    save("wikipedia", obs=obs, log_probs=em.logProbs, lt=th._log_transmat, pi=th._log_startprob,
         frame=frame, vit_logprob=lp, vit_path=np.asarray(path, dtype=np.int64), fwd=fwd, bwd=bwd,
         fwd_logprob=flp, post=post, twin_vit_logprob=lp_tw, twin_vit_path=np.asarray(path_tw),
         twin_post=post_tw, obs4=obs4, log_probs4=em4.logProbs, vit_logprob4=lp4,
         vit_path4=np.asarray(path4, dtype=np.int64))

    # ------------------------------------------------------------------ 2. array-level kernels
    shapes = [("n5", 5, (4, 4, 4), (), 1000, (0, 1, 2)),
              ("n35", 35, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, 1500, (0, 1)),
              ("n100", 100, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, 400, (0,))]
    for tag, N, syms, gauss, T, seeds in shapes:
        for seed in seeds:
            for with_ratio in (0, 1):
                model = synth.make_model(N, syms, gauss, seed=seed, sparse=0.3 if seed == 1 else 0.0)
                obs = synth.sample_obs(model, T, seed=seed + 100, missing=0.02)
                K, S = model.n_tracks, model.log_probs.shape[2]
                ratios = None
                if with_ratio:
                    # dpBenchmark.py:128-133 style U(0.01, 10) for seed 0, segment-like otherwise
                    if seed == 0:
                        random.seed(200)
                        ratios = np.asarray([random.uniform(0.01, 10.) for _ in range(T)])
                    else:
                        ratios = synth.random_ratios(T, seed=seed)
                lt, pi, lp_tab = model.log_transmat, model.log_startprob, model.log_probs
                frame = np.zeros((T, N))
                _emission.fastAllLogProbs(obs, lp_tab, frame, 1.0, ratios)
                frame_norm = np.zeros((T, N))
                _emission.fastAllLogProbs(obs, lp_tab, frame_norm, 3.0 / K, ratios)
                fwd = np.zeros((T, N))
                _hmm._forward(T, N, pi, lt, frame, ratios, fwd)
                bwd = np.zeros((T, N))
                _hmm._backward(T, N, pi, lt, frame, ratios, bwd)
                path, vlp = _hmm._viterbi(T, N, pi, lt, ratios, frame)
                lnP = logsumexp(fwd[-1])
                xi = np.zeros((N, N))
                _hmm._log_sum_lneta(T, N, fwd, lt, bwd, frame, lnP, ratios, xi)
                gamma = fwd + bwd
                post_fit = np.exp(gamma.T - logsumexp(gamma, axis=1)).T
                post_eval = post_fit + np.finfo(np.float32).eps
                post_eval /= np.sum(post_eval, axis=1).reshape((-1, 1))
                stats = np.zeros((K, N, S))
                _emission.fastAccumulateStats(obs, stats, post_fit, ratios)
                rows = rows_subset(T)
                full = (tag == "n5")
                d = dict(obs=obs, log_probs=lp_tab, lt=lt, pi=pi,
                         ratios=(ratios if ratios is not None else np.zeros(0)),
                         rows=rows, frame_rows=frame[rows], frame_norm_rows=frame_norm[rows],
                         frame_sha=np.frombuffer(hashlib.sha256(frame.tobytes()).digest(), dtype=np.uint8),
                         fwd_rows=fwd[rows], bwd_rows=bwd[rows], fwd_logprob=lnP,
                         vit_path=np.asarray(path, dtype=np.int16), vit_logprob=vlp,
                         xi_logsum=xi, post_fit_rows=post_fit[rows], post_eval_rows=post_eval[rows],
                         obs_stats=stats)
                if full:
                    d.update(frame=frame, fwd=fwd, bwd=bwd, post_eval=post_eval)
                save("kern_%s_s%d_r%d" % (tag, seed, with_ratio), **d)

    # ------------------------------------------------------------------ 3. quirk fixtures
    # Q9: leading impossible rows are zeroed, a later impossible row stays -inf.
    em = IndependentMultinomialEmissionModel(2, [2], [[[0.0, 1.0], [0.0, 1.0]]])
    obs = np.asarray([[1], [1], [2], [1], [2]], dtype=np.uint8)
    with np.errstate(divide="ignore"):
        frame = em.allLogProbs(obs)
    save("quirk_q9_leading_rows", obs=obs, log_probs=em.logProbs, frame=frame)

    # Q1: zero transitions become -1e100; forward lattice carries -1e100-ish cells.
    model = synth.make_model(6, (3, 4), (), seed=7, sparse=0.6)
    T = 300
    obs = synth.sample_obs(model, T, seed=8)
    hq, _ = make_ref_hmm(model)
    frame = hq._compute_log_likelihood(obs)
    lpq, fwdq = hq._do_forward_pass(frame)
    bwdq = hq._do_backward_pass(frame)
    vlpq, pathq = hq._do_viterbi_pass(frame)
    flpq, postq = hq.score_samples(obs)
    save("quirk_q1_zero_transitions", obs=obs, log_probs=model.log_probs, lt=hq._log_transmat,
         pi=hq._log_startprob, transmat=model.transmat, frame=frame, fwd=fwdq, bwd=bwdq,
         fwd_logprob=lpq, vit_path=np.asarray(pathq, dtype=np.int64), vit_logprob=vlpq, post=postq)

    # Q5/Q6: exact ties -- all log-probs are multiples of 0.5 so many candidates tie exactly.
    rs = np.random.RandomState(11)
    N, K, T = 7, 2, 400
    lt = -0.5 * rs.randint(1, 4, size=(N, N)).astype(np.float64)
    pi = -0.5 * rs.randint(1, 3, size=N).astype(np.float64)
    lp_tab = np.zeros((K, N, 4))
    lp_tab[:, :, 1:] = -0.5 * rs.randint(1, 4, size=(K, N, 3))
    obs = rs.randint(1, 4, size=(T, K)).astype(np.uint8)
    frame = np.zeros((T, N))
    _emission.fastAllLogProbs(obs, lp_tab, frame, 1.0, None)
    path, vlp = _hmm._viterbi(T, N, pi, lt, None, frame)
    ratios = 0.5 * rs.randint(1, 6, size=T).astype(np.float64)
    path_r, vlp_r = _hmm._viterbi(T, N, pi, lt, ratios, frame)
    save("quirk_ties", obs=obs, log_probs=lp_tab, lt=lt, pi=pi, frame=frame, ratios=ratios,
         vit_path=np.asarray(path, dtype=np.int64), vit_logprob=vlp,
         vit_path_r=np.asarray(path_r, dtype=np.int64), vit_logprob_r=vlp_r)

    # Q11/Q12/Q14: driver asymmetries on a segmented TrackTable (decode applies ratios to
    # transitions only; score_samples applies none; decode(algorithm="map") runs Viterbi).
    model = synth.make_model(5, (4, 4, 4), (), seed=3)
    T = 500
    obs = synth.sample_obs(model, T, seed=4)
    lens = np.minimum(1 + np.random.RandomState(5).geometric(1 / 20.0, size=T), 100)
    tab = segmented_table(obs, lens)
    hd, emd = make_ref_hmm(model, eff_len=20)
    ratios = emd.getSegmentRatios(tab)
    lp_dec, path_dec = hd.decode(tab)
    lp_map, path_map = hd.decode(tab, algorithm="map")
    flp_ss, post_ss = hd.score_samples(tab)
    frame_tab = hd._compute_log_likelihood(tab)        # TrackTable kept -> ratios applied
    save("driver_asymmetry", obs=obs, seg_lens=lens, eff_len=20, ratios=ratios,
         log_probs=model.log_probs, transmat=model.transmat, startprob=np.exp(model.log_startprob),
         lt=hd._log_transmat, pi=hd._log_startprob,
         decode_logprob=lp_dec, decode_path=np.asarray(path_dec, dtype=np.int64),
         map_logprob=lp_map, map_path=np.asarray(path_map, dtype=np.int64),
         score_logprob=flp_ss, score_post=post_ss, frame_with_ratios=frame_tab)

    # ------------------------------------------------------------------ 4. dpBenchmark frame
    def makeFrame(numStates, numObs):                    # restated formula of dpBenchmark.py:90-98
        frame = np.zeros((numObs, numStates))
        for i in range(numObs):
            for j in range(numStates):
                frame[i, j] = myLog(float(j) / float(numStates)) + myLog((float(i % 9) + 1.) / 10)
        return frame
    S_, N_ = 5, 100000
    mthmm = MultitrackHmm(emissionModel=IndependentMultinomialEmissionModel(S_, [2]))
    frame = makeFrame(S_, N_)
    vlp, vpath = mthmm._do_viterbi_pass(frame)
    flp, ftab = mthmm._do_forward_pass(frame)
    btab = mthmm._do_backward_pass(frame)
    random.seed(200)
    segr = np.asarray([random.uniform(0.01, 10.) for _ in range(N_)])
    mthmm.emissionModel.getSegmentRatios = lambda x: segr
    vlp_r, vpath_r = mthmm._do_viterbi_pass(frame)
    flp_r, ftab_r = mthmm._do_forward_pass(frame)
    btab_r = mthmm._do_backward_pass(frame)
    save("dpbenchmark_s5_n100000", lt=mthmm._log_transmat, pi=mthmm._log_startprob,
         vit_logprob=vlp, vit_path=np.asarray(vpath, dtype=np.uint8), fwd_logprob=flp,
         fwd_last=ftab[-1], bwd_first=btab[0],
         vit_logprob_r=vlp_r, vit_path_r=np.asarray(vpath_r, dtype=np.uint8), fwd_logprob_r=flp_r,
         fwd_last_r=ftab_r[-1], bwd_first_r=btab_r[0])

    # ------------------------------------------------------------------ 5. one EM iteration
    for with_ratio in (0, 1):
        model = synth.make_model(6, (3, 5, 4), (), seed=21)
        seqs, lens_l, tabs = [], [], []
        for i, T in enumerate((200, 1, 350)):
            o = synth.sample_obs(model, T, seed=30 + i, missing=0.03)
            ln = np.minimum(1 + np.random.RandomState(40 + i).geometric(1 / 20.0, size=T), 100)
            seqs.append(o)
            lens_l.append(ln)
            tabs.append(segmented_table(o, ln) if with_ratio else o)
        he, eme = make_ref_hmm(model, eff_len=(20 if with_ratio else None), n_iter=2, thresh=0.0,
                               fixStart=False, fudge=0.0)
        captured = {}
        orig_mstep = he._do_mstep

        def spy(stats, params, _orig=orig_mstep, _c=captured):
            if "start" not in _c:
                _c.update(start=stats["start"].copy(), trans=stats["trans"].copy(),
                          obs=stats["obs"].copy(), nobs=stats["nobs"])
            return _orig(stats, params)
        he._do_mstep = spy
        # init_params default would reset start/trans: keep ours (hmm.py:618-620 / basehmm.py:669-673)
        he.init_params = ""
        he.fit(tabs)
        save("em_iteration_r%d" % with_ratio,
             **{"obs%d" % i: s for i, s in enumerate(seqs)},
             **{"seg_lens%d" % i: s for i, s in enumerate(lens_l)},
             eff_len=20, log_probs=model.log_probs, transmat=model.transmat,
             startprob=np.exp(model.log_startprob), symbols=np.asarray(model.symbols_per_track),
             stats_start=captured["start"], stats_trans=captured["trans"], stats_obs=captured["obs"],
             stats_nobs=captured["nobs"], transmat_after=he.transmat_, startprob_after=he.startprob_,
             log_probs_after=he.emissionModel.logProbs, last_logprob=he.last_forward_log_prob)

    shutil.rmtree(root, ignore_errors=True)
    print("done; reference scratch build removed:", root)


if __name__ == "__main__":
    main()
