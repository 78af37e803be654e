#!/usr/bin/env python3
"""Timing of the N = 100 segmented configuration (BASELINE config 5 shape) through the generic
one-wave-per-interval kernels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nint, T = 64, 20000
model = synth.make_model(N, seed=1)
offs = np.arange(nint + 1, dtype=np.int64) * T
obs = synth.random_obs(model, nint * T, seed=2)
ratios = synth.random_ratios(nint * T, seed=3)
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, symbols_per_track=model.symbols_per_track)
hb = HipBatch(obs, offs, ratios)
for it in range(2):
    t0 = time.perf_counter()
    hm.eval(hb, viterbi=True, posterior=True)
    dt = time.perf_counter() - t0
    print("N=%d: %d intervals x %d: %.3f s -> %.3e positions/s  %s" % (N, nint, T, dt, nint * T / dt, hb.timing()))
