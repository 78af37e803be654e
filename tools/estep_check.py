#!/usr/bin/env python3
"""Development check of the fused (chunk-parallel) E-step against the CPU oracle and the sequential path:
python tools/estep_check.py [N] [n_intervals] [interval_len]"""
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 35
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
    model = synth.make_model(N, synth.CONFIG4_SYMBOLS, synth.CONFIG4_GAUSSIAN, seed=12)
    rs = np.random.RandomState(3)
    lens = [L + int(rs.randint(-L // 3, L // 3)) for _ in range(n)] + [1, 70, 1500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
    K, _, S = model.log_probs.shape
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    out = {}
    for mode in ("1", "0"):
        os.environ["TEHMM_ESTEP_FUSED"] = mode
        hb = HipBatch(obs, offs)
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp = hm.estep(hb, False, start, trans, st)
        t0 = time.perf_counter()
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp = hm.estep(hb, False, start, trans, st)
        dt = time.perf_counter() - t0
        out[mode] = (lp, start, trans, st, hb.interval_logprobs())
        print("fused=%s  %.2f ms  lp %.9e  %s" % (mode, dt * 1e3, lp, hb.timing()))
        hb.close()
    ref = oracle.estep([obs[offs[i]:offs[i + 1]] for i in range(len(lens))], model.log_probs, model.log_startprob,
                       model.log_transmat, 1.0, None)

    def rel(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        m = np.abs(b) > 1e-9
        return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0
    for mode in ("1", "0"):
        lp, start, trans, st, ilp = out[mode]
        print("fused=%s vs oracle: lp %.3g start %.3g trans %.3g obs %.3g | abs: start %.3g trans %.3g obs %.3g" % (
            mode, abs(lp - ref["logprob"]) / abs(ref["logprob"]), rel(start, ref["start"]), rel(trans, ref["trans"]),
            rel(st, ref["obs"]), np.abs(start - ref["start"]).max(), np.abs(trans - ref["trans"]).max(),
            np.abs(st - ref["obs"]).max()))


if __name__ == "__main__":
    main()
