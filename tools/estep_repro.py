#!/usr/bin/env python3
"""Same input -> same bits: the fused E-step twice (and on a second batch handle) must give bit-identical statistics.
python tools/estep_repro.py [N] [Mb]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tehmm_amd import synth
from tehmm_amd.engine import HipBatch, HipModel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 35
mb = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
model = synth.make_model(N, synth.CONFIG4_SYMBOLS, synth.CONFIG4_GAUSSIAN, seed=12)
lens = synth.interval_lengths(int(mb * 1e6), 100_000, 100_000, seed=3)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
obs = synth.sample_obs(model, int(offs[-1]), seed=5, missing=0.02)
K, _, S = model.log_probs.shape
hm = HipModel(model.log_transmat, model.log_startprob, model.log_probs, 1.0, model.symbols_per_track)
runs = []
for rep in range(3):
    hb = HipBatch(obs, offs)
    for again in range(2):
        start, trans, st = np.zeros(N), np.zeros((N, N)), np.zeros((K, N, S))
        lp = hm.estep(hb, False, start, trans, st)
        runs.append((lp, start.copy(), trans.copy(), st.copy()))
    hb.close()
same = all(r[0] == runs[0][0] and np.array_equal(r[1], runs[0][1]) and np.array_equal(r[2], runs[0][2]) and
           np.array_equal(r[3], runs[0][3]) for r in runs[1:])
print("runs", len(runs), "logprob", runs[0][0], "bit-identical:", same)
if not same:
    for i, r in enumerate(runs[1:], 1):
        print(i, np.abs(r[2] - runs[0][2]).max(), np.abs(r[3] - runs[0][3]).max(), np.abs(r[1] - runs[0][1]).max(), r[0] - runs[0][0])
sys.exit(0 if same else 1)
