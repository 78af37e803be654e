"""E-step on the item-parallel passes (tehmm_wide_estep.hip.h) against the oracle: N >= 64 and / or segment ratios.
usage: python tools/wide_estep_check.py [N ...]        (TEHMM_HIP_LIB selects a development build)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle                                   # noqa: E402
from tehmm_amd import synth                                 # noqa: E402
from tehmm_amd.engine import HipBatch, HipModel             # noqa: E402


def rel(a, b, floor=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.abs(b) > floor
    return float(np.max(np.abs(a - b)[m] / np.abs(b)[m])) if m.any() else 0.0


def ratios_for(total, seed, mean=4.0):
    rs = np.random.RandomState(seed)
    r = np.clip(rs.geometric(1.0 / mean, size=total), 1, 60).astype(np.float64) / mean
    r[rs.rand(total) < 0.3] = 1.0
    return r


def run(N, symbols, gauss, use_ratios, lens, seed=5):
    model = synth.make_model(N, symbols, gauss, seed=12 + N)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    obs = synth.sample_obs(model, int(offs[-1]), seed=seed, missing=0.02)
    K, _, S = model.log_probs.shape
    r = ratios_for(int(offs[-1]), seed + 1) if use_ratios else None
    hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
    hb = HipBatch(obs, offs, r)
    start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
    lp = hm.estep(hb, use_ratios, start, trans, st)
    t1 = time.perf_counter()
    s2, t2, st2 = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
    lp2 = hm.estep(hb, use_ratios, s2, t2, st2)
    dt = time.perf_counter() - t1
    tm = hb.timing()
    hb.close()
    hm.close()
    t0 = time.perf_counter()
    ref = oracle.estep([obs[offs[i]:offs[i + 1]] for i in range(len(lens))], model.log_probs, model.log_startprob,
                       model.log_transmat, 1.0,
                       [r[offs[i]:offs[i + 1]] for i in range(len(lens))] if use_ratios else None)
    to = time.perf_counter() - t0
    errs = dict(lp=abs(lp - ref["logprob"]) / abs(ref["logprob"]), start=rel(start, ref["start"]),
                trans=rel(trans, ref["trans"], 1e-6), obs=rel(st, ref["obs"], 1e-6))
    same = lp == lp2 and np.array_equal(trans, t2) and np.array_equal(st, st2) and np.array_equal(start, s2)
    ok = max(errs.values()) <= 1e-6 and "estep_reduce" in tm and "estep_emission_rows" in tm
    print("N=%d ratios=%s total=%d: %s  errs %s  reproducible %s  second call %.1f ms (oracle %.1f s)  timing %s" % (
        N, use_ratios, int(offs[-1]), "PASS" if ok and same else "FAIL",
        {k: float("%.3g" % v) for k, v in errs.items()}, same, dt * 1e3, to,
        {k: round(v, 2) for k, v in tm.items()}), flush=True)
    return ok and same


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [100, 35]
    rs = np.random.RandomState(1)
    good = True
    for N in sizes:
        lens = [int(x) for x in rs.randint(3000, 9000, size=4)] + [1, 70, 1500, 1024, 2048 + 64]
        sym = (3, 5, 4, 30) if N >= 64 else (3, 5, 4, 30, 250)
        gs = () if N >= 64 else (4,)
        for use_r in ((False, True) if N >= 64 else (True,)):
            good &= run(N, sym, gs, use_r, lens)
    print("RESULT", "PASS" if good else "FAIL")
    sys.exit(0 if good else 1)
