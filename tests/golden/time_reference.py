#!/usr/bin/env python3
"""Times the real reference (cythonized in a scratch dir, see make_golden.py) against the CPU oracle on
identical inputs in THIS container: the ratio BASELINE.md asks for (`oracle_over_restatement`), i.e. proof
that the oracle used as bench.py's cpu_baseline is not a strawman.  Re-run: python tests/golden/time_reference.py"""
import os, shutil, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402


def main():
    root = mg.build_reference()
    from teHmm import _hmm, _emission
    from teHmm.basehmm import logsumexp
    from tehmm_amd import synth
    from oracle import oracle
    model = synth.make_model(35, seed=0)
    T = 200_000
    obs = synth.sample_obs(model, T, seed=1)
    lt, pi, lp = model.log_transmat, model.log_startprob, model.log_probs
    N = 35
    best = {}
    for rep in range(3):
        t0 = time.perf_counter()
        frame = np.zeros((T, N)); _emission.fastAllLogProbs(obs, lp, frame, 1.0, None)
        fwd = np.zeros((T, N)); _hmm._forward(T, N, pi, lt, frame, None, fwd)
        bwd = np.zeros((T, N)); _hmm._backward(T, N, pi, lt, frame, None, bwd)
        gamma = fwd + bwd
        post = np.exp(gamma.T - logsumexp(gamma, axis=1)).T
        post += np.finfo(np.float32).eps; post /= post.sum(axis=1).reshape((-1, 1))
        frame2 = np.zeros((T, N)); _emission.fastAllLogProbs(obs, lp, frame2, 1.0, None)
        path, vlp = _hmm._viterbi(T, N, pi, lt, None, frame2)
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        flp_o, post_o = oracle.score_samples(obs, lp, pi, lt)
        vlp_o, path_o = oracle.decode(obs, lp, pi, lt)
        t_or = time.perf_counter() - t0
        best["ref"] = min(best.get("ref", 1e9), t_ref)
        best["oracle"] = min(best.get("oracle", 1e9), t_or)
        assert np.array_equal(path, path_o) and vlp == vlp_o
        np.testing.assert_allclose(post, post_o, rtol=1e-9)
    print("reference (Cython, 1 core) %.2f s = %.0f positions/s; oracle (C, 1 core) %.2f s = %.0f positions/s; "
          "oracle_over_restatement (reference time / oracle time) = %.2f"
          % (best["ref"], T / best["ref"], best["oracle"], T / best["oracle"], best["ref"] / best["oracle"]))
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
