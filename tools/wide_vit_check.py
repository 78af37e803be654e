#!/usr/bin/env python3
"""Development check of the chunk-parallel exact Viterbi for 64 <= N <= 128 (tehmm_wide.hip.h) against the CPU oracle:
[SPARSE=0.5] [STAY=0.995] [SEED=0] python tools/wide_vit_check.py [N] [n_intervals] [interval_len] [ratio 0/1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tehmm_amd import synth
    from tehmm_amd.engine import HipBatch, HipModel
    from oracle import oracle
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    with_ratio = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    kw = {}
    if os.environ.get("SPARSE"):
        kw["sparse"] = float(os.environ["SPARSE"])
    if os.environ.get("STAY"):
        kw["stay"] = float(os.environ["STAY"])
    model = synth.make_model(N, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=int(os.environ.get("SEED", 0)), **kw)
    rs = np.random.RandomState(3)
    lens = [L + int(rs.randint(-L // 3, L // 3)) for _ in range(n)] + [1, 70, 1500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(offs[-1])
    obs = synth.sample_obs(model, T, seed=5, missing=0.02)
    ratios = None
    if with_ratio:
        ratios = synth.random_ratios(T, seed=7)
        ratios[rs.rand(T) < 0.4] = 1.0
        ratios = np.ascontiguousarray(ratios)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    out = {}
    for mode in ("1", "0"):
        os.environ["TEHMM_WIDE_VIT"] = mode
        hb = HipBatch(obs, offs, ratios)
        res = hm.eval(hb, viterbi=True, posterior=False, use_ratios=with_ratio)
        t0 = time.perf_counter()
        res = hm.eval(hb, viterbi=True, posterior=False, use_ratios=with_ratio)
        dt = time.perf_counter() - t0
        out[mode] = (res["viterbi_logprob"].copy(), np.array(hb.paths()))
        print("wide_vit=%s  %.2f ms  %s" % (mode, dt * 1e3, hb.timing()), flush=True)
        hb.close()
    ok = True
    for i in range(len(lens)):
        a, b = int(offs[i]), int(offs[i + 1])
        lp_o, path_o = oracle.decode(obs[a:b], model.log_probs, model.log_startprob, model.log_transmat, 1.0,
                                     None if ratios is None else ratios[a:b])
        for mode in ("1", "0"):
            lp, paths = out[mode]
            same = np.array_equal(paths[a:b], path_o)
            if not same or lp[i] != lp_o:
                ok = False
                nd = int(np.sum(paths[a:b] != path_o))
                first = int(np.argmax(paths[a:b] != path_o)) if nd else -1
                print("interval %d (len %d) wide_vit=%s: path diffs %d (first at %d)  score %r vs %r" % (
                    i, b - a, mode, nd, first, lp[i], lp_o))
    print("ALL EXACT" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
