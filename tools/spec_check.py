#!/usr/bin/env python3
"""Bit-exactness check of the chunk-parallel Viterbi against the CPU oracle on long intervals, with
jump statistics (diagnostic)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tehmm_amd import synth, _lib
from tehmm_amd.engine import HipBatch, HipModel
from oracle import oracle

def main():
    N = int(os.environ.get("N", "35"))
    lens = [int(float(x)) for x in (sys.argv[1:] or ["300000", "150000", "5000", "700001"])]
    model = synth.make_model(N, seed=0)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    piece = synth.sample_obs(model, 50000, seed=11)
    total = int(offs[-1])
    obs = np.tile(piece, (total // 50000 + 1, 1))[:total].copy()
    rs = np.random.RandomState(5)
    noise = rs.rand(total) < 0.3
    for k, sk in enumerate(model.symbols_per_track):
        obs[noise, k] = rs.randint(1, sk + 1, size=int(noise.sum()))
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
    hb = HipBatch(obs, offs)
    for it in range(2):
        t0 = time.perf_counter()
        res = hm.eval(hb, viterbi=True, posterior=False)
        dt = time.perf_counter() - t0
    print("GPU eval %.1f ms  timing %s" % (dt * 1e3, {k: round(v, 2) for k, v in hb.timing().items()}))
    paths = hb.paths()
    bad = 0
    for i in range(len(lens)):
        sl = slice(offs[i], offs[i + 1])
        lp, p = oracle.decode(obs[sl], model.log_probs, model.log_startprob, model.log_transmat)
        same = np.array_equal(paths[sl], p)
        print("interval %d T=%d: path %s, logprob gpu=%r cpu=%r %s" % (
            i, lens[i], "EXACT" if same else "MISMATCH at %d" % int(np.argmax(paths[sl] != p)),
            res["viterbi_logprob"][i], lp, "ok" if res["viterbi_logprob"][i] == lp else "DIFF"))
        bad += (not same) or res["viterbi_logprob"][i] != lp
    print("RESULT", "FAIL" if bad else "PASS")
    return bad

if __name__ == "__main__":
    sys.exit(int(main()))
