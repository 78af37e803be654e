#!/usr/bin/env python3
"""Development check of the chunk-parallel posterior for 64 <= N <= 128 (tehmm_wide.hip.h) against the CPU oracle:
python tools/wide_check.py [N] [n_intervals] [interval_len]"""
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
    model = synth.make_model(N, synth.CONFIG2_SYMBOLS, synth.CONFIG2_GAUSSIAN, seed=0)
    rs = np.random.RandomState(3)
    lens = [L + int(rs.randint(-L // 3, L // 3)) for _ in range(n)] + [1, 70, 1500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    out = {}
    for mode in ("1", "0"):
        os.environ["TEHMM_WIDE_CP"] = mode
        hb = HipBatch(obs, offs)
        res = hm.eval(hb, viterbi=False, posterior=True)
        t0 = time.perf_counter()
        res = hm.eval(hb, viterbi=False, posterior=True)
        dt = time.perf_counter() - t0
        out[mode] = (res["forward_logprob"].copy(), np.array(hb.posteriors()))
        print("wide_cp=%s  %.2f ms  %s" % (mode, dt * 1e3, hb.timing()), flush=True)
        hb.close()
    p_o, vlp_o, flp_o, post_o = oracle.eval_batch(obs, offs, model.log_probs, model.log_startprob, model.log_transmat,
                                                  1.0, None, n_threads=8)
    for mode in ("1", "0"):
        flp, post = out[mode]
        print("wide_cp=%s vs oracle: logprob rel %.3g  posterior max rel %.3g  max abs %.3g" % (
            mode, np.max(np.abs(flp - flp_o) / np.abs(flp_o)), np.max(np.abs(post - post_o) / post_o),
            np.max(np.abs(post - post_o))))


if __name__ == "__main__":
    main()
